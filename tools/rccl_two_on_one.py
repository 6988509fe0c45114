#!/usr/bin/env python3
"""Two RCCL ranks sharing the ONE GPU of the box: does `dist.gather_packed` move bytes between two processes
on hardware at least once?  RCCL, like NCCL, normally refuses two ranks on one device ("Duplicate GPU detected"); this script
tries anyway, with the packed images of two real shards from Program.match_device_packed, bounded by a timeout in the parent.

    python tools/rccl_two_on_one.py            parent: starts ranks 0 and 1, waits at most 120 s, prints a verdict line (JSON)
"""
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import faulthandler
    faulthandler.enable()
    import torch
    import torch.distributed as dist
    import forgex_amd
    from forgex_amd import dist as fxdist, synth
    rank, world = int(os.environ["RANK"]), 2
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n_total, L = 200_000 + 37, 128   # uneven shards
    a, b = fxdist.shard_bounds(n_total, rank, world)
    rows = synth.batch("cfg5", a, b - a, dev)
    prog = forgex_amd.Program(synth.PATTERNS["cfg5"], forgex_amd.OP_SEARCH)
    packed = prog.match_device_packed(rows, spans=True)
    bufs = fxdist.gather_buffers(n_total, L, True, dev)
    res = fxdist.gather_packed(packed, n_total, L, True, buffers=bufs)
    ok = None
    if rank == 0:
        shards, sizes = res
        # the other rank's shard, recomputed here: what arrived over RCCL must be what that rank's kernel produced
        rows1 = synth.batch("cfg5", sizes[0], sizes[1], dev)
        want1 = prog.match_device_packed(rows1, spans=True)
        torch.cuda.synchronize()
        ok = bool(torch.equal(shards[0], packed[:shards[0].numel()]) and torch.equal(shards[1], want1[:shards[1].numel()]))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"two_ranks_one_gpu": "ok" if ok else "MISMATCH", "shard_rows": sizes}), flush=True)


def main():
    if "RANK" in os.environ:
        return child()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    t0 = time.time()
    verdict = None
    while time.time() - t0 < 120 and any(p.poll() is None for p in procs):
        time.sleep(0.5)
    timed_out = any(p.poll() is None for p in procs)
    for p in procs:
        if p.poll() is None:
            p.kill()
    outs = [p.communicate() for p in procs]
    for o, e in outs:
        for ln in o.decode().splitlines():
            if ln.startswith("{"):
                verdict = json.loads(ln)
    if verdict is None:
        err = (outs[0][1].decode() + outs[1][1].decode())
        key = [ln for ln in err.splitlines() if "uplicate" in ln or "Error" in ln or "error" in ln][:4]
        verdict = {"two_ranks_one_gpu": "refused" if not timed_out else "timeout", "rc": [p.returncode for p in procs], "stderr_key_lines": key}
    print(json.dumps(verdict))


if __name__ == "__main__":
    main()
