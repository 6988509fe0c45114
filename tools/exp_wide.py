import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time, os
import forgex_amd
from forgex_amd import synth
dev=torch.device("cuda")
rows3=synth.batch("cfg3",0,10_000_000,dev)
rows4=synth.batch("cfg4",0,1048576,dev)
def run(pat, rows, label):
    p=forgex_amd.Program(pat, forgex_amd.OP_SEARCH)
    p.match_device(rows); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(10): p.match_device(rows)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    print("%-8s %-34s path %d  %.3f ms  %.0f GB/s"%(label, pat, p.last_path(), dt*1e3, rows.numel()/dt/1e9), flush=True)
for wide in (True, False):
    if wide: os.environ.pop("FXAMD_NO_W16", None)
    else: os.environ["FXAMD_NO_W16"]="1"
    forgex_amd.lib().fxamd_reload_env()   # (the hooks are read once per process)
    lab="wide" if wide else "chain"
    for pat in ("\\d{3}-\\d{4}", "\\w+@\\w+\\.[a-z]+", "[a-z]{3,5}\\d{2,4}x", "(19|20)\\d\\d-(0[1-9]|1[012])"):
        run(pat, rows3, lab)
    run("[α-ωぁ-ん]+", rows4, lab)
    run("[ぁ-ん]+[0-9]*", rows4, lab)
