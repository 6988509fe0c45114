#!/usr/bin/env python3
"""Which resource is the cold-clock penalty of the headline kernel?

Per-launch durations (HIP events around every launch) of three things on the config-3 batch, each after three different
histories, in ONE process on ONE box:

    what   = copy        plain device-to-device copy of the 2.56 GB batch (torch copy_: read + write, no compute)
             kernel      fx_search_fast first pass alone (fxamd_launch_fast_only) of the library in FXAMD_LIB
                         (run the script once with the product library and once with the FX_EXP_NOCOMPUTE build)
    before = idle        0.5 s of sleep
             generator   the batch generated again on the GPU (seconds of ALU-heavy torch kernels) -- the driver's protocol
             stream      100 back-to-back device copies (a memory-bound history)

While the launches run, a sampler thread reads the sysfs clock tables (pp_dpm_sclk / pp_dpm_mclk / pp_dpm_fclk: the starred
level) about every millisecond, so that the per-launch curve can be laid next to what the clocks did.

Output: one JSON document on stdout.  Usage (GPU box):  FXAMD_LIB=... python tools/exp_transient.py [--launches 60]
"""
import argparse
import glob
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def find_clock_files():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk"):
            p = os.path.join(card, name)
            if os.path.exists(p) and name not in out:
                out[name] = p
    return out


def starred(path):
    try:
        for ln in open(path).read().splitlines():
            if ln.rstrip().endswith("*"):
                return ln.split(":")[1].strip().rstrip("*").strip()
    except Exception:
        return None
    return None


class Sampler(threading.Thread):
    def __init__(self, files):
        super().__init__(daemon=True)
        self.files, self.samples, self.stop_flag = files, [], False

    def run(self):
        while not self.stop_flag:
            t = time.perf_counter()
            self.samples.append((t,) + tuple(starred(p) for p in self.files.values()))
            time.sleep(0.001)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=60)
    ap.add_argument("--rows", type=int, default=10_000_000)
    args = ap.parse_args()
    import torch
    import forgex_amd
    from forgex_amd import synth
    dev = torch.device("cuda", 0)
    n, L = args.rows, 256
    rows = synth.batch("cfg3", 0, n, dev)
    prog = forgex_amd.Program(synth.PATTERNS["cfg3"], forgex_amd.OP_SEARCH)
    flags = torch.empty(n, dtype=torch.uint8, device=dev)
    frm = torch.empty(n, dtype=torch.int32, device=dev)
    to = torch.empty(n, dtype=torch.int32, device=dev)
    scratch = torch.empty_like(rows)
    lib = forgex_amd.lib()
    stream = torch.cuda.current_stream(dev)

    def launch_kernel():
        rc = lib.fxamd_launch_fast_only(prog._h, rows.data_ptr(), n, L, flags.data_ptr(), frm.data_ptr(), to.data_ptr(), stream.cuda_stream)
        assert rc == 0, rc

    def launch_copy():
        scratch.copy_(rows)

    whats = {"copy": launch_copy, "kernel": launch_kernel}

    def before_idle():
        time.sleep(0.5)

    def before_generator():
        nonlocal rows
        r2 = synth.batch("cfg3", 0, n, dev)   # (not synchronised: the launches follow the generator kernels in the stream)
        rows = r2

    def before_stream():
        for _ in range(100):
            scratch.copy_(rows)

    befores = {"idle": before_idle, "generator": before_generator, "stream": before_stream}
    files = find_clock_files()
    res = {"lib": os.environ.get("FXAMD_LIB", "forgex_amd/libforgex_amd.so"), "launches": args.launches, "clock_files": files, "runs": []}
    launch_kernel()
    launch_copy()
    torch.cuda.synchronize()
    for rep in range(2):
        for wname, what in whats.items():
            for bname, before in befores.items():
                torch.cuda.synchronize()
                time.sleep(0.2)
                smp = Sampler(files)
                before()
                smp.start()
                t0 = time.perf_counter()
                evs = []
                for _ in range(args.launches):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(stream)
                    what()
                    b.record(stream)
                    evs.append((a, b))
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                smp.stop_flag = True
                smp.join()
                ms = [a.elapsed_time(b) for a, b in evs]
                clocks = {}
                for i, name in enumerate(files):
                    seq = [s[1 + i] for s in smp.samples]
                    # run-length summary: (value, first sample time in ms since the first launch was enqueued)
                    rl = []
                    for s, v in zip(smp.samples, seq):
                        if not rl or rl[-1][0] != v:
                            rl.append((v, round((s[0] - t0) * 1e3, 2)))
                    clocks[name] = rl[:40]
                k = args.launches
                res["runs"].append({"rep": rep, "what": wname, "before": bname, "wall_ms": (t1 - t0) * 1e3,
                                    "first5_ms": sum(ms[:5]) / 5, "l5_25_ms": sum(ms[5:25]) / 20, "last20_ms": sum(ms[k - 20:]) / 20,
                                    "per_launch_ms": [round(x, 4) for x in ms], "clocks": clocks})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
