#!/usr/bin/env python3
"""bench.py -- input GB/s scanned by the batch `.in.`/regex hot path on MI355X.

A "step" is ONE pass of the hot path over the resident batch: fxamd_match_batch_device (flags + (from,to) spans
for every row) on BASELINE.json config 3 -- `[a-z]+\\d+` over 10M x 256 B synthetic rows, inputs already in HBM.
With --gpus N each rank owns its own 10M-row shard of an N*10M-row batch (weak scaling, no data-path collective);
the packed-result gather over RCCL is timed separately and reported as `gather_ms`.

One JSON line on rank 0:
  value        whole-job input GB/s = N * rows * row_len * steps / max-over-ranks wall time
  roofline     dominant kernel (fx_search_fast) vs the HBM roofline: algorithmic bytes per launch
               (rows * (row_len + 9): input once + 1 flag + two int32) / its average launch duration, measured
               live with HIP events on the launch stream
  cpu_baseline the REAL reference (oracle/_ref/ref_driver, flang build; kind "reference") or the C++ restatement
               (oracle/liboracle.so; kind "port") on a bounded sample of the same rows, on this host's cores
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SETTLE = 30              # untimed launches before any timed leg, so that the clocks have settled (see the warm-up comment below)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 achievable


def cpu_baseline(cfg, pattern, row_len, budget_s=15.0):
    """Reference CPU path on a bounded sample of the SAME workload rows (rank 0, N=1 only)."""
    import numpy as np
    import torch
    from forgex_amd import synth
    threads = os.cpu_count() or 1
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    use_ref = os.path.exists(ref) and os.access(ref, os.X_OK)

    def run(nrows):
        rows = synth.batch(cfg, 0, nrows, torch.device("cpu")).numpy()
        if use_ref:
            with tempfile.NamedTemporaryFile(suffix=".rows", delete=False) as f:
                f.write(rows.tobytes())
                path = f.name
            try:
                line = "B R %s %d %d %s - %d\n" % (pattern.encode().hex().upper(), row_len, nrows, path, threads)
                out = subprocess.run([ref], input=line.encode(), capture_output=True, timeout=600).stdout.decode().split()
            finally:
                os.unlink(path)
            if len(out) < 2 or out[0] != "B":
                raise RuntimeError("ref_driver: " + " ".join(out))
            return float(out[1])
        sys.path.insert(0, os.path.join(ROOT, "tests", "support"))
        import oracle_lib
        t0 = time.perf_counter()
        oracle_lib.batch(2, pattern.encode(), rows, threads)
        return time.perf_counter() - t0

    def table_walker():
        """Second, stronger CPU baseline: the product's own compiled tables walked on ONE host core by the
        test harness (tests/support/libhostwalk.so: one compile per batch, linear-time passes) -- reported, never a fallback."""
        path = os.path.join(ROOT, "tests", "support", "libhostwalk.so")
        if not os.path.exists(path):
            return None
        lib = ctypes.CDLL(path)
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        lib.hw_batch.argtypes = [ctypes.c_char_p, i64, ctypes.c_int, vp, i64, i64, vp, vp, vp]
        nrows = 200000
        rows = np.ascontiguousarray(synth.batch(cfg, 0, nrows, torch.device("cpu")).numpy())
        f = np.zeros(nrows, np.uint8)
        a = np.zeros(nrows, np.int32)
        b = np.zeros(nrows, np.int32)
        pat = pattern.encode()
        t0 = time.perf_counter()
        st = lib.hw_batch(pat, len(pat), 0, rows.ctypes.data_as(vp), nrows, row_len, f.ctypes.data_as(vp), a.ctypes.data_as(vp), b.ctypes.data_as(vp))
        dt = time.perf_counter() - t0
        if st != 0:
            return None
        return {"value": nrows * row_len / dt / 1e9, "unit": "GB/s", "cores": 1, "kind": "product tables on the host (test harness)",
                "sample": "first %d rows of %s, %.2f s wall, one compile per batch" % (nrows, cfg, dt)}

    try:
        walker = table_walker()
    except Exception as e:
        walker = {"value": None, "sample": "failed: %r" % (e,)}
    try:
        probe = 4 * threads
        t = run(probe)
        per_row = max(t / probe, 1e-9)
        sample = int(min(max(probe, budget_s / per_row), 200000))
        sample = max(threads, (sample // threads) * threads)
        t = run(sample)
        return {"value": sample * row_len / t / 1e9, "unit": "GB/s", "cores": threads,
                "kind": "reference" if use_ref else "port",
                "sample": "first %d rows of %s (%d B each), %.1f s wall, per-row compile as the elemental operator does" % (
                    sample, cfg, row_len, t),
                "us_per_row": t / sample * 1e6 * 1.0, "table_walker": walker}
    except Exception as e:   # the baseline is reported, never allowed to sink the bench line
        return {"value": None, "unit": "GB/s", "cores": threads, "kind": "reference" if use_ref else "port", "sample": "failed: %r" % (e,)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30,
                    help="untimed steps first; the clocks settle over the first ~20 back-to-back launches after an idle gap (DESIGN.md 4.1)")
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: the config's size; cfg5: 12.5M)")
    ap.add_argument("--flags-only", action="store_true", help="time the flags-only `.in.` entry instead of flags+spans")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import forgex_amd
    from forgex_amd import synth
    from forgex_amd import dist as fxdist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    assert torch.cuda.is_available(), "bench.py needs a GPU: the match path has no CPU fallback"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    cfg = args.config
    n_cfg, row_len = synth.SHAPES[cfg]
    rows_per_gpu = args.rows or (n_cfg if cfg != "cfg5" else n_cfg // 8)
    pattern = synth.PATTERNS[cfg]
    start = rank * rows_per_gpu
    rows = synth.batch(cfg, start, rows_per_gpu, dev)
    prog = forgex_amd.Program(pattern, forgex_amd.OP_SEARCH)
    assert prog.status == 0
    spans = not args.flags_only
    flags = torch.empty(rows_per_gpu, dtype=torch.uint8, device=dev)
    frm = torch.empty(rows_per_gpu, dtype=torch.int32, device=dev) if spans else None
    to = torch.empty(rows_per_gpu, dtype=torch.int32, device=dev) if spans else None
    out = (flags, frm, to)

    def step():
        prog.match_device(rows, spans=spans, out=out)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # clock settling: after an idle gap the first ~20 back-to-back launches run at drifting clocks (0.53 -> 0.67 -> 0.50 ms per
    # launch on config 3); when fewer than SETTLE warm-up steps were asked for, the difference is run first, untimed,
    # and reported as config.clock_settle_steps
    settle = max(0, SETTLE - args.warmup)
    for _ in range(settle):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    n_matches = int((flags != 0).sum().item())

    # ---- roofline leg: the dominant kernel alone, HIP events on its launch stream --------------------------
    L = forgex_amd.lib()
    stream = torch.cuda.current_stream(dev)
    reps = max(5, min(args.steps, 200))
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    kernel_ms = None
    fast = prog.last_path() in (1, 3, 5, 6, 7, 8)
    whole_step = cfg == "cfg4"   # non-ASCII rows: the work is in the SECOND pass (on-device UTF-8 decode + scan) -> time the whole step
    if fast:   # the CPU-side work since the timed region left the GPU idle: settle the clocks again, as the warm-up steps did
        for _ in range(max(args.warmup, SETTLE)):
            step()
    if fast and whole_step:
        for a, b in evs:
            a.record(stream)
            step()
            b.record(stream)
        torch.cuda.synchronize()
        kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / reps
    elif fast:
        for a, b in evs:
            a.record(stream)
            rc = L.fxamd_launch_fast_only(prog._h, rows.data_ptr(), rows_per_gpu, row_len, flags.data_ptr(),
                                          frm.data_ptr() if spans else None, to.data_ptr() if spans else None, stream.cuda_stream)
            assert rc == 0, rc
            b.record(stream)
        torch.cuda.synchronize()
        kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / reps
        step()   # restore complete results (fix-up pass) before the gather below
        torch.cuda.synchronize()
    out_bytes = 9 if spans else 1
    alg_bytes = rows_per_gpu * (row_len + out_bytes)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(cfg)
        except Exception:
            traffic = None
    kname = "fx_search_fast<%d>" % (row_len // 16)
    if whole_step:
        kname += " first pass + second pass (UTF-8 decode in LDS + scan)"
    roofline = {"bound": "hbm", "kernel": kname if fast else "fx_general",
                "achieved": (alg_bytes / (kernel_ms * 1e-3) / 1e9) if kernel_ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (alg_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if kernel_ms else None,
                "traffic": traffic, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes}

    # ---- the pure `.in.` verdict (flags only, no spans) on the same rows: an extra, separately timed leg ---------------------------
    flags_only = None
    if spans:
        try:
            out_f = (flags, None, None)
            for _ in range(max(SETTLE, args.warmup)):
                prog.match_device(rows, spans=False, out=out_f)
            barrier()
            f0 = time.perf_counter()
            for _ in range(args.steps):
                prog.match_device(rows, spans=False, out=out_f)
            torch.cuda.synchronize()
            fdt = time.perf_counter() - f0
            flags_only = {"value": world * rows_per_gpu * row_len * args.steps / fdt / 1e9, "unit": "GB/s (this rank's time, all ranks' bytes)",
                          "ms_per_step": fdt / args.steps * 1e3}
            step()   # restore flags + spans for the checks below
            torch.cuda.synchronize()
        except Exception:
            flags_only = None

    # ---- measured device-copy ceiling in the same run: rows -> scratch, read + write bytes per second ----
    copy_gbs = None
    try:
        scratch = torch.empty_like(rows)
        for _ in range(max(SETTLE, args.warmup)):
            scratch.copy_(rows)
        ncopy = max(5, min(args.steps, 50))
        ca, cb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ca.record(stream)
        for _ in range(ncopy):
            scratch.copy_(rows)
        cb.record(stream)
        torch.cuda.synchronize()
        copy_gbs = 2.0 * rows.numel() * ncopy / (ca.elapsed_time(cb) * 1e-3) / 1e9
        del scratch
    except Exception:
        copy_gbs = None
    roofline["device_copy_gbs"] = copy_gbs   # bytes read + bytes written per second of a plain device-to-device copy of the batch

    # ---- packed result gather over RCCL (not part of `value`) ---------------------------------------------------
    gather_ms = None
    if world > 1:
        f2 = flags
        a2 = frm if spans else torch.zeros(rows_per_gpu, dtype=torch.int32, device=dev)
        b2 = to if spans else torch.zeros(rows_per_gpu, dtype=torch.int32, device=dev)
        fxdist.gather_results(f2, a2, b2, rows_per_gpu * world, row_len)   # warm-up (RCCL connection setup)
        barrier()
        g0 = time.perf_counter()
        res = fxdist.gather_results(f2, a2, b2, rows_per_gpu * world, row_len)
        barrier()
        gather_ms = (time.perf_counter() - g0) * 1e3
        if rank == 0:
            assert res[0].shape[0] == rows_per_gpu * world

    if rank == 0:
        total_bytes = world * rows_per_gpu * row_len * args.steps
        line = {
            "metric": "input GB/s scanned (.in. over 10M strings)", "value": total_bytes / dt / 1e9, "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: `%s` .in. (flags%s) over %d x %d B rows per GPU, counter-based PRNG rows resident in HBM" % (
                cfg, pattern, " + (from,to) spans" if spans else " only", rows_per_gpu, row_len),
                "rows_per_gpu": rows_per_gpu, "row_len": row_len, "pattern": pattern, "parallelism": "shard%d" % world,
                "outputs": "flag u8 + from/to int32" if spans else "flag u8", "matches_rank0": n_matches,
                "clock_settle_steps": settle},
            "frac_of_hbm_peak": total_bytes / dt / 1e9 / (HBM_PEAK_GBS * world),
            "frac_of_one_eighth_gpu": total_bytes / dt / 1e9 / (HBM_PEAK_GBS / 8 * world),
            "roofline": roofline, "gather_ms": gather_ms, "flags_only": flags_only,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, pattern, row_len)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
