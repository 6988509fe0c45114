#!/usr/bin/env python3
"""bench.py -- input GB/s scanned by the batch `.in.`/regex hot path on MI355X.

A "step" is ONE pass of the hot path over the resident batch: fxamd_match_batch_device (flags + (from,to) spans
for every row) on BASELINE.json config 3 -- `[a-z]+\\d+` over 10M x 256 B synthetic rows, inputs already in HBM.
With --gpus N > 1 the headline workload is what BASELINE.json's north star words: "a 10M-string synthetic batch at 1, 2, 4 and 8 GPUs" --
config 3's 10M rows split into N contiguous shards, rank i = rows [i*N/W, (i+1)*N/W) (`"scaling": "strong"`, the default for config 3 since
round 5; no data-path collective); the same run also times N x 10M rows (every rank its own config-sized shard) and reports it as
`weak_scaling_extra`.  `--scaling weak` makes that the headline instead (the default of rounds 1-4), `--rows R` fixes the rows per rank.
The packed-result gather over RCCL is timed separately and reported as `gather_ms`.

Documented multi-GPU invocations (the driver's `--gpus N --steps K --warmup W` form is the first):
    python bench.py --gpus 8                         config 3's 10M rows split over 8 GPUs (+ weak_scaling_extra: 8 x 10M rows)
    python bench.py --gpus 8 --scaling weak          weak scaling on config 3: 8 x 10M x 256 B as the headline
    python bench.py --gpus 8 --config cfg5           BASELINE config 5: 100M x 128 B sharded over 8 GPUs + RCCL gather of the spans

Launch: `python bench.py --gpus N` starts the N rank processes itself (one per GPU, RCCL rendezvous on 127.0.0.1) when no
launcher did; under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the launcher's
RANK / LOCAL_RANK / WORLD_SIZE are used.  The parent never touches the GPU: it only relays rank 0's JSON line.

One JSON line on rank 0:
  value        whole-job input GB/s = N * rows * row_len * steps / max-over-ranks wall time of EXACTLY `--steps` steps after
               EXACTLY `--warmup` untimed ones (so a short run carries the clock transient of the first launches after an
               idle gap); `settled` repeats the measurement after SETTLE more launches
  roofline     dominant kernel vs the HBM roofline: algorithmic bytes per launch (rows * (row_len + 9): input once +
               1 flag + two int32) / its average launch duration, measured live with HIP events on the launch stream,
               at settled clocks (`cold_*`: the same over the first launches after an idle gap)
  parity       GPU flags / from / to of THIS run against the product's tables walked on the host (test harness
               tests/support/libhostwalk.so, all host cores) over the WHOLE batch of the rank; `parity.oracle`: the same batch against
               oracle/liboracle.so (the reference's algorithm restated: no matching code shared with the product) -- as many rows as
               about 20 s of this host's cores allow, spread over the batch (the whole of config 3 on the 256-core GPU box);
               multi-rank runs: `parity.gathered_shards` = every other rank's gathered + unpacked results against the host walker
  cpu_baseline the REAL reference (oracle/_ref/ref_driver, flang build; kind "reference") or the C++ restatement
               (oracle/liboracle.so; kind "port") on a bounded sample of the same rows, on this host's cores, and the GPU
               results compared with the reference's on that sample
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SETTLE = 30              # back-to-back launches after which the clocks have settled
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 achievable
DRYRUN = os.environ.get("FXAMD_BENCH_DRYRUN") == "1"   # CPU plumbing check of the N-rank path (gloo, no GPU, no timing claims)


def spawn_ranks(args, argv):
    """--gpus N without a launcher: start N children (RANK / LOCAL_RANK / WORLD_SIZE set), relay rank 0's line.  Runs BEFORE
    anything touches the GPU; the parent stays a plain Python process."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = subprocess.PIPE if r == 0 else sys.stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out))
    out0 = procs[0].communicate()[0].decode()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def host_walker():
    path = os.path.join(ROOT, "tests", "support", "libhostwalk.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.hw_batch_mt.argtypes = [ctypes.c_char_p, i64, ctypes.c_int, vp, i64, i64, vp, vp, vp, ctypes.c_int]
    return lib


def full_parity(pattern, rows_dev, flags, frm, to, threads):
    """The rank's WHOLE batch: GPU results vs the product's tables walked on the host.  Chunked so that the host
    copy of the rows stays bounded."""
    import numpy as np
    lib = host_walker()
    if lib is None:
        return {"rows": 0, "mismatches": None, "checker": "host table walker (tests/support/libhostwalk.so missing)"}
    vp = ctypes.c_void_p
    n, L = rows_dev.shape
    pat = pattern.encode()
    bad = 0
    first_bad = None
    t0 = time.perf_counter()
    step = max(1, (1 << 30) // max(L, 1))
    for c0 in range(0, n, step):
        c1 = min(n, c0 + step)
        rows = np.ascontiguousarray(rows_dev[c0:c1].cpu().numpy())
        m = c1 - c0
        f = np.zeros(m, np.uint8)
        a = np.zeros(m, np.int32)
        b = np.zeros(m, np.int32)
        st = lib.hw_batch_mt(pat, len(pat), 0, rows.ctypes.data_as(vp), m, L, f.ctypes.data_as(vp), a.ctypes.data_as(vp), b.ctypes.data_as(vp), threads)
        if st != 0:
            return {"rows": 0, "mismatches": None, "checker": "host table walker: compile status %d" % st}
        d = f != flags[c0:c1].cpu().numpy()
        if frm is not None:
            d |= (a != frm[c0:c1].cpu().numpy()) | (b != to[c0:c1].cpu().numpy())
        k = int(d.sum())
        if k and first_bad is None:
            first_bad = int(c0 + np.flatnonzero(d)[0])
        bad += k
    return {"rows": int(n), "mismatches": bad, "first_mismatch_row": first_bad, "fields": "flag, from, to" if frm is not None else "flag",
            "checker": "product tables walked on the host (tests/support/libhostwalk.so, %d threads), whole batch of rank 0" % threads,
            "seconds": round(time.perf_counter() - t0, 2)}


def oracle_parity(pattern, rows_dev, flags, frm, to, threads, budget_s=20.0):
    """The timed batch against the ORACLE (oracle/liboracle.so: the C++ restatement of the reference's restart-loop algorithm, pinned
    to the real reference by the golden vectors) -- a checker that shares no matching code with the product, unlike the host table
    walker of full_parity.  As many rows as fit `budget_s` on this host's cores (a probe sets the rate), taken as 16 slices spread
    evenly over the batch; on the 256-core GPU box that is the whole of config 3."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "support"))
    import oracle_lib
    n, L = rows_dev.shape
    pat = pattern.encode()
    probe = min(n, 64 * threads)
    t0 = time.perf_counter()
    oracle_lib.batch(2, pat, rows_dev[:probe].cpu().numpy(), threads)
    per_row = max((time.perf_counter() - t0) / probe, 1e-9)
    want = int(min(n, max(probe, budget_s / per_row)))
    slices = 16 if want < n else 1
    per = max(1, want // slices)
    bad = checked = 0
    first_bad = None
    t0 = time.perf_counter()
    for i in range(slices):
        c0 = 0 if slices == 1 else (n - per) * i // max(1, slices - 1)
        c1 = n if slices == 1 else min(n, c0 + per)
        rows = np.ascontiguousarray(rows_dev[c0:c1].cpu().numpy())
        f, a, b = oracle_lib.batch(2, pat, rows, threads)
        d = f != flags[c0:c1].cpu().numpy()
        if frm is not None:
            d |= (a != frm[c0:c1].cpu().numpy()) | (b != to[c0:c1].cpu().numpy())
        k = int(d.sum())
        if k and first_bad is None:
            first_bad = int(c0 + np.flatnonzero(d)[0])
        bad += k
        checked += c1 - c0
    return {"rows": int(checked), "of": int(n), "mismatches": bad, "first_mismatch_row": first_bad, "slices": slices,
            "checker": "oracle/liboracle.so (restatement of api_internal_m.F90:31-167, %d threads)" % threads,
            "seconds": round(time.perf_counter() - t0, 2)}


def cpu_baseline(cfg, pattern, row_len, gpu_results, budget_s=15.0):
    """Reference CPU path on a bounded sample of the SAME workload rows (rank 0, N=1 only); its outputs double as a checker of the
    GPU results on that sample."""
    import numpy as np
    import torch
    from forgex_amd import synth
    threads = os.cpu_count() or 1
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    use_ref = os.path.exists(ref) and os.access(ref, os.X_OK)

    def run(nrows, want_out=False):
        rows = synth.batch(cfg, 0, nrows, torch.device("cpu")).numpy()
        if use_ref:
            with tempfile.NamedTemporaryFile(suffix=".rows", delete=False) as f:
                f.write(rows.tobytes())
                path = f.name
            opath = path + ".out" if want_out else "-"
            try:
                line = "B R %s %d %d %s %s %d\n" % (pattern.encode().hex().upper(), row_len, nrows, path, opath, threads)
                out = subprocess.run([ref], input=line.encode(), capture_output=True, timeout=600).stdout.decode().split()
                res = np.loadtxt(opath, dtype=np.int64).reshape(-1, 3) if want_out else None
            finally:
                os.unlink(path)
                if want_out and os.path.exists(opath):
                    os.unlink(opath)
            if len(out) < 2 or out[0] != "B":
                raise RuntimeError("ref_driver: " + " ".join(out))
            return float(out[1]), res
        sys.path.insert(0, os.path.join(ROOT, "tests", "support"))
        import oracle_lib
        t0 = time.perf_counter()
        f, a, b = oracle_lib.batch(2, pattern.encode(), rows, threads)
        return time.perf_counter() - t0, np.stack([f.astype(np.int64), a.astype(np.int64), b.astype(np.int64)], axis=1)

    def table_walker():
        """Second, stronger CPU baseline: the product's own compiled tables walked on ONE host core by the
        test harness (tests/support/libhostwalk.so: one compile per batch, linear-time passes) -- reported, never a fallback."""
        lib = host_walker()
        if lib is None:
            return None
        vp = ctypes.c_void_p
        nrows = 200000
        rows = np.ascontiguousarray(synth.batch(cfg, 0, nrows, torch.device("cpu")).numpy())
        f = np.zeros(nrows, np.uint8)
        a = np.zeros(nrows, np.int32)
        b = np.zeros(nrows, np.int32)
        pat = pattern.encode()
        t0 = time.perf_counter()
        st = lib.hw_batch_mt(pat, len(pat), 0, rows.ctypes.data_as(vp), nrows, row_len, f.ctypes.data_as(vp), a.ctypes.data_as(vp), b.ctypes.data_as(vp), 1)
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
        t, _ = run(probe)
        per_row = max(t / probe, 1e-9)
        sample = int(min(max(probe, budget_s / per_row), 200000))
        sample = max(threads, (sample // threads) * threads)
        t, res = run(sample, want_out=True)
        check = None
        if res is not None and gpu_results is not None:
            gf, ga, gb = gpu_results
            k = min(sample, gf.shape[0])
            bad = int(((res[:k, 0] != gf[:k]) | (res[:k, 1] != ga[:k]) | (res[:k, 2] != gb[:k])).sum())
            check = {"rows": k, "mismatches": bad, "checker": "the real reference (oracle/_ref/ref_driver)" if use_ref else "oracle/liboracle.so"}
        return {"value": sample * row_len / t / 1e9, "unit": "GB/s", "cores": threads,
                "kind": "reference" if use_ref else "port",
                "sample": "first %d rows of %s (%d B each), %.1f s wall, per-row compile as the elemental operator does" % (
                    sample, cfg, row_len, t),
                "us_per_row": t / sample * 1e6 * 1.0, "gpu_vs_reference_on_sample": check, "table_walker": walker}
    except Exception as e:   # the baseline is reported, never allowed to sink the bench line
        return {"value": None, "unit": "GB/s", "cores": threads, "kind": "reference" if use_ref else "port", "sample": "failed: %r" % (e,)}


def host_path_rate(fx, prog, cfg, row_len, nrows=1 << 20):
    """PCIe-inclusive rate of the host-buffer entry the Fortran module uses (fxamd_match_batch_host: pageable caller memory in,
    results out), on a bounded slice of the workload.  Reported next to `value`, never as `value`."""
    import numpy as np
    import torch
    from forgex_amd import synth
    rows = np.ascontiguousarray(synth.batch(cfg, 0, nrows, torch.device("cpu")).numpy())
    prog.match_host(rows, spans=True)   # first call allocates the handle's chunk slots
    reps = 3

    def rate():
        t0 = time.perf_counter()
        for _ in range(reps):
            res = prog.match_host(rows, spans=True)
        return (time.perf_counter() - t0) / reps, res
    dt, res = rate()
    out = {"value": nrows * row_len / dt / 1e9, "unit": "GB/s of input, PCIe-inclusive (H2D rows + kernels + D2H results), pageable caller memory",
           "rows": nrows, "ms_per_call": dt * 1e3, "matches": int(res[0].sum())}
    try:   # the same with the caller's array pinned in place (fxamd_host_register): the rows travel by DMA from the caller's memory
        with fx.pinned(rows):
            dtp, _ = rate()
        out["pinned"] = {"value": nrows * row_len / dtp / 1e9, "ms_per_call": dtp * 1e3}
    except Exception as e:
        out["pinned"] = {"value": None, "error": repr(e)}
    return out


def dryrun(args, rank, world, emit):
    """FXAMD_BENCH_DRYRUN=1: the N-rank plumbing on CPU (gloo): rendezvous, shard bounds, packed gather, max-over-ranks reduction and
    the one JSON line -- no GPU, no match calls, no throughput claim."""
    import torch
    import torch.distributed as dist
    from forgex_amd import dist as fxdist
    from forgex_amd import synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = args.config
    _, row_len = synth.SHAPES[cfg]
    rows_per_gpu = args.rows or 4096
    start = rank * rows_per_gpu
    idx = torch.arange(start, start + rows_per_gpu)
    flags = (idx % 3 == 0).to(torch.uint8)
    frm = ((idx % row_len) + 1).to(torch.int32) * flags
    to = torch.full_like(frm, row_len) * flags
    census = fxdist.job_census(torch.device("cpu"))
    per_rank = fxdist.gather_floats(0.001 * (rank + 1), torch.device("cpu"))
    tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    res = fxdist.gather_results(flags, frm, to, rows_per_gpu * world, row_len)
    ok = None
    if rank == 0:
        all_idx = torch.arange(0, rows_per_gpu * world)
        ef = (all_idx % 3 == 0).to(torch.uint8)
        ok = bool(torch.equal(res[0], ef) and torch.equal(res[1], ((all_idx % row_len) + 1).to(torch.int32) * ef))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        emit({"metric": "input GB/s scanned (.in. over 10M strings)", "value": None, "unit": "GB/s", "n_gpus": world, "dryrun": True,
              "ranks_joined": int(tt.item()), "gather_ok": ok, "rccl_ranks": census["ranks"], "devices": census["devices"],
              "devices_distinct": census["distinct"], "collective_backend": census["backend"], "per_rank_ms_per_step": [x * 1e3 for x in per_rank], "steps": args.steps, "warmup": args.warmup, "scaling": args.scaling,
              "config": {"workload": "dry run of the %d-rank plumbing on CPU (gloo); no GPU work" % world, "parallelism": "shard%d" % world,
                         "rows_per_gpu": rows_per_gpu}})
    return 0


def single_process(args, emit):
    """--single-process: ONE process drives the N GPUs of the node, and every shard's packed results land in the ROOT GPU's buffer while the shard is
    scanned -- the collective-free delivery SURVEY.md section 5 names ("7 peer writes into the root's buffer").  Peer access is enabled from every
    device to device 0; the root's buffer holds one packed image per shard (forgex_amd.dist.peer_direct_layout); shard i's match call gets its slice
    of that buffer as `d_packed` (the C ABI takes any device pointer), so GPU i's kernels store their flag words and narrow spans over xGMI as they
    go.  There is no gather step: the step's time already contains the delivery.  One host thread per GPU enqueues that GPU's launches (a 1.25 M-row
    shard is a ~66 us step: one thread enqueueing for eight devices would be the bottleneck), each on its own program handle (handles from blobs are
    never shared: no common mutex).  Same timing contract as the multi-rank path: W untimed steps, then K timed ones between two barriers of all
    threads with every device synchronised, the slowest device's time counts.
    FXAMD_BENCH_DRYRUN=1: the same plumbing on the CPU (threads, shard bounds, image offsets, unpack of the root's buffer) without a GPU."""
    import threading
    import numpy as np
    import torch
    from forgex_amd import synth
    from forgex_amd import dist as fxdist
    N = args.gpus
    cfg = args.config
    n_cfg, row_len = synth.SHAPES[cfg]
    spans = not args.flags_only
    scaling = args.scaling or ("strong" if cfg == "cfg3" and not args.rows else "weak")
    if scaling == "strong" and not args.rows:
        bounds = [fxdist.shard_bounds(n_cfg, r, N) for r in range(N)]
    else:
        per = args.rows or (n_cfg if cfg != "cfg5" else n_cfg // 8)
        bounds = [(r * per, (r + 1) * per) for r in range(N)]
    if DRYRUN:
        bounds = [(a, a + min(b - a, args.rows or 4096)) for a, b in bounds]
    sizes = [b - a for a, b in bounds]
    offs, total = fxdist.peer_direct_layout(sizes, row_len, spans)
    pattern = synth.PATTERNS[cfg]
    root_dev = torch.device("cpu") if DRYRUN else torch.device("cuda", 0)
    root = torch.zeros(max(total, 16), dtype=torch.uint8, device=root_dev)
    peer = []
    if not DRYRUN:
        import forgex_amd
        assert torch.cuda.is_available() and torch.cuda.device_count() >= N, "--single-process --gpus %d: %d devices visible" % (N, torch.cuda.device_count())
        hip = ctypes.CDLL("libamdhip64.so")
        for d in range(1, N):   # device d may write device 0's memory (error 704 = already enabled)
            assert hip.hipSetDevice(d) == 0
            rc = hip.hipDeviceEnablePeerAccess(0, 0)
            peer.append({"device": d, "to": 0, "torch_can_access": bool(torch.cuda.can_device_access_peer(d, 0)), "hipDeviceEnablePeerAccess": int(rc)})
            assert rc in (0, 704), "no peer access from device %d to device 0 (hip error %d): peer-direct delivery needs xGMI / PCIe peer access" % (d, rc)
        hip.hipSetDevice(0)
        base = forgex_amd.Program(pattern, forgex_amd.OP_SEARCH)
        assert base.status == 0
        blob = base.blob()
    bar = threading.Barrier(N + 1)
    state = {"dt": [0.0] * N, "dt_settled": [0.0] * N, "err": [None] * N, "path": [None] * N, "init_ms": [0.0] * N}
    rows_keep = [None] * N

    def worker(d):
        try:
            a, b = bounds[d]
            m = b - a
            image = root[offs[d]:offs[d] + max(fxdist._packed_total(m, row_len, spans), 16)]
            if DRYRUN:
                idx = torch.arange(a, b)
                f = (idx % 3 == 0).to(torch.uint8)
                fr = ((idx % row_len) + 1).to(torch.int32) * f
                to = torch.full_like(fr, row_len) * f

                def step():
                    image.copy_(fxdist.pack_image(f, fr, to, row_len, spans))
                sync = lambda: None
            else:
                dev = torch.device("cuda", d)
                torch.cuda.set_device(dev)
                prog = forgex_amd.Program.from_blob(blob, forgex_amd.OP_SEARCH)   # (its own handle: no mutex shared with the other devices' threads)
                t_i = time.perf_counter()
                init_rows = synth.batch(cfg, a, 64, dev)
                prog.match_device_packed(init_rows, spans=spans)
                assert forgex_amd.lib().fxamd_program_reserve(prog._h, m, torch.cuda.current_stream(dev).cuda_stream) == 0
                torch.cuda.synchronize(dev)
                state["init_ms"][d] = (time.perf_counter() - t_i) * 1e3
                rows = synth.batch(cfg, a, m, dev)
                rows_keep[d] = rows

                def step():
                    prog.match_device_packed(rows, spans=spans, out=image)
                sync = lambda: torch.cuda.synchronize(dev)
            for _ in range(args.warmup):
                step()
            for key in ("dt", "dt_settled"):
                sync()
                bar.wait()            # every device idle, every thread here: the timed region starts
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                sync()
                state[key][d] = time.perf_counter() - t0
                bar.wait()            # ... and ends when the slowest device has finished
                if key == "dt":
                    for _ in range(SETTLE if not DRYRUN else 0):
                        step()
            if not DRYRUN:
                state["path"][d] = prog.last_path()
        except Exception as e:   # a failed thread must not leave the others in a barrier
            state["err"][d] = repr(e)
            bar.abort()

    threads = [threading.Thread(target=worker, args=(d,)) for d in range(N)]
    for t in threads:
        t.start()
    wall = {}
    try:
        for key in ("dt", "dt_settled"):
            bar.wait()
            t0 = time.perf_counter()
            bar.wait()
            wall[key] = time.perf_counter() - t0
    except threading.BrokenBarrierError:
        pass
    for t in threads:
        t.join()
    assert not any(state["err"]), state["err"]
    rows_all = sum(sizes)
    dt = wall["dt"]
    # ---- what arrived in the root's buffer: every shard's image unpacked ON THE ROOT and compared (generator truth in the dry run; the host table
    #      walker over the regenerated rows on GPUs) -- the peer writes are inside what is checked ----
    shards = []
    for d in range(N):
        a, b = bounds[d]
        m = b - a
        image = root[offs[d]:offs[d] + max(fxdist._packed_total(m, row_len, spans), 16)]
        if DRYRUN:
            f, fr, to = fxdist.unpack_image(image, m, row_len, spans)
            idx = torch.arange(a, b)
            ef = (idx % 3 == 0).to(torch.uint8)
            ok = bool(torch.equal(f, ef) and (not spans or torch.equal(fr, ((idx % row_len) + 1).to(torch.int32) * ef)))
            shards.append({"device": d, "rows": m, "mismatches": 0 if ok else 1})
        elif not args.no_parity:
            f, fr, to = forgex_amd.unpack_results(image, m, row_len, spans)   # (on device 0: the root's copy)
            torch.cuda.synchronize()
            res = full_parity(pattern, rows_keep[d], f, fr, to, os.cpu_count() or 1)
            shards.append({"device": d, "rows": m, "mismatches": res["mismatches"], "matches": int(f.sum().item())})
    devices = [("cpu-thread-%d" % d) if DRYRUN else fxdist.device_identity(torch.device("cuda", d)) for d in range(N)]
    per_image = [max(fxdist._packed_total(m, row_len, spans), 16) for m in sizes]
    emit({"metric": "input GB/s scanned (.in. over 10M strings)", "value": None if DRYRUN else rows_all * row_len * args.steps / dt / 1e9, "unit": "GB/s", "n_gpus": N,
          "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u8",
          "data": "synthetic", "dryrun": DRYRUN or None, "single_process": True,
          "settled": {"value": None if DRYRUN else rows_all * row_len * args.steps / wall["dt_settled"] / 1e9, "ms_per_step": wall["dt_settled"] / args.steps * 1e3},
          "per_device_ms_per_step": [x / args.steps * 1e3 for x in state["dt"]], "per_device_ms_per_step_settled": [x / args.steps * 1e3 for x in state["dt_settled"]],
          "devices": devices, "devices_distinct": len({repr(x) for x in devices}) == N, "peer_access": peer, "init_ms": state["init_ms"], "kernel_path": state["path"],
          "delivery": {"mode": "peer-direct: every shard's kernels write their packed image into the root GPU's buffer (device 0) while they scan; no collective, no gather step",
                       "bytes_per_row": per_image[0] / max(sizes[0], 1), "bytes_into_root_over_links_per_step": int(sum(per_image[1:])), "image_offsets": offs, "root_buffer_bytes": total},
          "parity": {"root_buffer_shards": shards, "mismatches": sum(x["mismatches"] or 0 for x in shards) if shards else None,
                     "checker": "generator truth" if DRYRUN else "product tables walked on the host over every shard's regenerated rows vs the root's unpacked images"},
          "config": {"workload": ("dry run of the single-process plumbing on CPU threads; no GPU work" if DRYRUN else
                                  "BASELINE %s: `%s` .in. over %d x %d B rows in %d contiguous shards, one process, packed results delivered peer-direct" % (cfg, pattern, rows_all, row_len, N)),
                     "parallelism": "shard%d" % N, "rows_per_gpu": sizes, "results": "packed (1 bit + 2 narrow spans per row)" if spans else "packed flag bits"}})
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--oracle-seconds", type=float, default=20.0,
                    help="time budget of the oracle leg of the whole-batch parity check (oracle/liboracle.so over as many rows as fit; ~70 s cover all of config 3 on the GPU box's 256 cores)")
    ap.add_argument("--warmup", type=int, default=30,
                    help="untimed steps first; the clocks settle over the first ~20-30 back-to-back launches after an idle gap (DESIGN.md 4.1)")
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: the config's size; cfg5: 12.5M)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak: every GPU scans its own config-sized shard; strong: the config's rows are split over the GPUs "
                         "(default: strong for config 3 at --gpus > 1 -- the north star's 10M-string batch at 1, 2, 4, 8 GPUs --, else weak)")
    ap.add_argument("--flags-only", action="store_true", help="time the flags-only `.in.` entry instead of flags+spans")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the whole-batch host check (profiling runs)")
    ap.add_argument("--no-extras", action="store_true", help="only the timed region and the roofline leg (profiling runs)")
    ap.add_argument("--single-process", action="store_true",
                    help="one process drives all --gpus N devices; every shard's packed results are written peer-direct into device 0's buffer (no collective)")
    args = ap.parse_args()
    if args.single_process:
        real_stdout = os.dup(1)
        os.dup2(2, 1)   # (libraries' banners go to stderr; the one JSON line to the real stdout)

        def emit1(obj):
            sys.stdout.flush()
            os.write(real_stdout, (json.dumps(obj) + "\n").encode())
        sys.exit(single_process(args, emit1))

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    scaling_defaulted = args.scaling is None
    if scaling_defaulted:
        args.scaling = "strong" if (world > 1 and args.config == "cfg3" and not args.rows) else "weak"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # stdout carries ONE line: everything a library prints there while the run lasts (RCCL prints a version banner on stdout) goes
    # to stderr instead, and rank 0's JSON line is written to the real stdout at the very end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    if DRYRUN:
        sys.exit(dryrun(args, rank, world, emit))

    import torch
    import torch.distributed as dist
    import forgex_amd
    from forgex_amd import synth
    from forgex_amd import dist as fxdist

    use_dist = world > 1 or os.environ.get("FXAMD_BENCH_FORCE_DIST") == "1"   # (world 1 + FORCE: the RCCL calls on a 1-GPU box)
    assert torch.cuda.is_available(), "bench.py needs a GPU: the match path has no CPU fallback"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    census = None
    if use_dist:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        # who takes part, as RCCL sees it (not as WORLD_SIZE claims): an all-reduce of 1 and an all-gather of every rank's device identity
        census = fxdist.job_census(dev)
        assert census["ranks"] == world, census
        assert census["distinct"] or world == 1, "two ranks share one GPU: %r" % (census["devices"],)

    cfg = args.config
    n_cfg, row_len = synth.SHAPES[cfg]
    if args.scaling == "strong" and not args.rows:
        # the config's rows (config 5: all 100M) split into contiguous shards, rank i = [i*N/W, (i+1)*N/W)
        start, stop = fxdist.shard_bounds(n_cfg, rank, world)
        rows_per_gpu = stop - start
    else:
        rows_per_gpu = args.rows or (n_cfg if cfg != "cfg5" else n_cfg // 8)
        start = rank * rows_per_gpu
    pattern = synth.PATTERNS[cfg]
    prog = forgex_amd.Program(pattern, forgex_amd.OP_SEARCH)
    assert prog.status == 0
    spans = not args.flags_only
    flags = torch.empty(rows_per_gpu, dtype=torch.uint8, device=dev)
    frm = torch.empty(rows_per_gpu, dtype=torch.int32, device=dev) if spans else None
    to = torch.empty(rows_per_gpu, dtype=torch.int32, device=dev) if spans else None
    out = (flags, frm, to)
    # Program initialisation, as a service does it once at start-up -- NOT a warm-up step of the workload: tables uploaded, the
    # stream's scratch reserved, and the kernels' code objects loaded by one call on 64 rows of the generator (HIP loads a code
    # object at the first launch of one of its kernels: tens of milliseconds of host work during which the GPU idles).  Measured
    # (profiles/r03_transient.md): the first ~40 launches after an idle GPU run up to 38 % slower -- the core clock, not memory: a
    # plain copy and the kernel's no-compute build show no such transient -- so an init that happens between the generator and the
    # warm-up steps puts the whole `--warmup 5 --steps 20` region inside that transient.
    init_rows = synth.batch(cfg, start, 64, dev)
    init_out = (torch.empty(64, dtype=torch.uint8, device=dev), torch.empty(64, dtype=torch.int32, device=dev) if spans else None,
                torch.empty(64, dtype=torch.int32, device=dev) if spans else None)
    t_init = time.perf_counter()
    prog.match_device(init_rows, spans=spans, out=init_out)
    rc = forgex_amd.lib().fxamd_program_reserve(prog._h, rows_per_gpu, torch.cuda.current_stream(dev).cuda_stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
    init_ms = (time.perf_counter() - t_init) * 1e3
    # the batch is generated on the GPU LAST (seconds of generator kernels), so that the warm-up steps follow a busy GPU, not an idle gap
    rows = synth.batch(cfg, start, rows_per_gpu, dev)

    def step():
        prog.match_device(rows, spans=spans, out=out)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    def timed(k):
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        barrier()
        dt = time.perf_counter() - t0
        per_rank = [dt]
        if use_dist:
            per_rank = fxdist.gather_floats(dt, dev)   # every rank's own time; the job's time is the slowest rank's
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        timed.per_rank = per_rank
        return dt

    # ---- the contract's region: exactly W untimed steps, then exactly K timed ones, max over ranks ------------------------------
    for _ in range(args.warmup):
        step()
    dt = timed(args.steps)
    per_rank_ms = [x / args.steps * 1e3 for x in timed.per_rank]
    n_matches = int((flags != 0).sum().item())
    rows_all = rows_per_gpu * world
    if use_dist and args.scaling == "strong":
        tt = torch.tensor([rows_per_gpu], dtype=torch.int64, device=dev)
        dist.all_reduce(tt)
        rows_all = int(tt.item())
    total_bytes = rows_all * row_len * args.steps
    # ---- the same again at settled clocks (SETTLE more back-to-back launches first) ------------------------------------------------
    for _ in range(SETTLE):
        step()
    dt_settled = timed(args.steps)

    # ---- multi-GPU runs of the north star's split batch: the weak-scaling figure (every rank its own config-sized shard) as a named extra ----
    weak_extra = None
    if world > 1 and args.scaling == "strong" and scaling_defaulted:
        rows_w = synth.batch(cfg, rank * n_cfg, n_cfg, dev)
        out_w = (torch.empty(n_cfg, dtype=torch.uint8, device=dev), torch.empty(n_cfg, dtype=torch.int32, device=dev) if spans else None,
                 torch.empty(n_cfg, dtype=torch.int32, device=dev) if spans else None)
        main_step = step

        def step():   # noqa: F811 (timed() is a closure of main() too: it calls whatever `step` is bound to in this scope)
            prog.match_device(rows_w, spans=spans, out=out_w)
        for _ in range(SETTLE):
            step()
        dt_w = timed(args.steps)
        weak_extra = {"value": world * n_cfg * row_len * args.steps / dt_w / 1e9, "unit": "GB/s", "ms_per_step": dt_w / args.steps * 1e3, "rows_per_gpu": n_cfg,
                      "rows_total": world * n_cfg, "scaling": "weak", "note": "every rank scans its own %d-row shard of a %d-times larger batch; same steps, settled clocks" % (n_cfg, world)}
        step = main_step
        del rows_w, out_w
        step()   # (the handle's results of the headline shard again)
        torch.cuda.synchronize()

    # ---- roofline leg: the dominant kernel alone, HIP events on its launch stream --------------------------------------------------
    L = forgex_amd.lib()
    stream = torch.cuda.current_stream(dev)
    reps = max(5, min(args.steps, 200))
    one_launch = prog.last_path() in (9, 10, 11, 12, 13, 14)   # fx_search_one: the step IS one kernel launch
    fast = one_launch or prog.last_path() in (1, 3, 5, 6, 7, 8, 16, 18)
    whole_step = one_launch or cfg == "cfg4"   # multi-pass pipeline on non-ASCII rows: several passes share the work -> time the whole step

    def kernel_events(k, group=1):
        """average duration of k launches; one HIP event pair around every `group` consecutive launches (an event costs about a microsecond of
        stream time: around EVERY launch that overstates kernels of tens of microseconds -- config 2: 21.0 us against rocprofv3's 18.8)"""
        k = (k + group - 1) // group * group
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k // group)]
        for a, b in evs:
            a.record(stream)
            for _ in range(group):
                if whole_step:
                    step()
                else:
                    rc = L.fxamd_launch_fast_only(prog._h, rows.data_ptr(), rows_per_gpu, row_len, flags.data_ptr(),
                                                  frm.data_ptr() if spans else None, to.data_ptr() if spans else None, stream.cuda_stream)
                    assert rc == 0, rc
            b.record(stream)
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in evs) / k

    kernel_ms = cold_ms = kernel_1 = None
    if fast:
        time.sleep(0.5)   # an idle gap, then the first launches: the cold figure
        cold_ms = kernel_events(min(reps, 20))
        for _ in range(SETTLE):
            step()
        kernel_1 = kernel_events(reps)             # an event pair around every launch
        kernel_ms = kernel_events(reps, group=10)   # an event pair around every ten launches: the figure of the roofline object
        step()   # restore complete results (later passes) before the checks below
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
    if prog.last_path() == 16:   # 256- / 128-byte rows: first pass (the timed kernel; half-row staging when spans are asked for) + one gated follow-up
        kname = ("fx_search_fast<%d, true, 0, 0, false, true, false> (half-row staging)" % (row_len // 32)) if spans else "fx_search_fast<16, false, 0, 0, false, false, false>"
    elif prog.last_path() == 18:   # the span kernel (rows of 128 / 64 / 32 / 16 bytes: a lane owns 128 bytes of whole rows) + one gated follow-up
        kname = "fx_search_span<%d, 0>" % row_len
    else:
        kname = ("fx_search_one<%d>" if one_launch else "fx_search_fast<%d>") % (row_len // 16)
    if whole_step and not one_launch:
        kname += ", all passes of one step"

    def gbs(ms):
        return (alg_bytes / (ms * 1e-3) / 1e9) if ms else None
    roofline = {"bound": "hbm", "kernel": kname if fast else "fx_general",
                "achieved": gbs(kernel_ms), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (gbs(kernel_ms) / HBM_PEAK_GBS) if kernel_ms else None,
                "traffic": traffic, "traffic_source": "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (committed, not measured in this run)" if traffic else None,
                "kernel_ms": kernel_ms, "kernel_ms_method": "HIP events on the launch stream, one pair per 10 consecutive launches, settled clocks",
                "kernel_ms_event_pair_per_launch": kernel_1 if fast else None, "algorithmic_bytes_per_launch": alg_bytes,
                "cold_kernel_ms": cold_ms, "cold_frac": (gbs(cold_ms) / HBM_PEAK_GBS) if cold_ms else None,
                "cold_note": "first %d launches after a 0.5 s idle gap" % min(reps, 20)}

    flags_only = copy_gbs = host_path = None
    if not args.no_extras:
        # ---- the pure `.in.` verdict (flags only, no spans) on the same rows: an extra, separately timed leg ---------------------------
        if spans:
            try:
                out_f = (flags, None, None)
                for _ in range(SETTLE):
                    prog.match_device(rows, spans=False, out=out_f)
                torch.cuda.synchronize()
                f0 = time.perf_counter()
                for _ in range(args.steps):
                    prog.match_device(rows, spans=False, out=out_f)
                torch.cuda.synchronize()
                fdt = time.perf_counter() - f0
                flags_only = {"value": rows_all * row_len * args.steps / fdt / 1e9, "unit": "GB/s (this rank's time, all ranks' bytes)",
                              "ms_per_step": fdt / args.steps * 1e3}
                step()   # restore flags + spans for the checks below
                torch.cuda.synchronize()
            except Exception:
                flags_only = None
        # ---- measured device-copy ceiling in the same run: rows -> scratch, read + write bytes per second ----
        try:
            scratch = torch.empty_like(rows)
            for _ in range(SETTLE):
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
        if rank == 0:
            try:
                host_path = host_path_rate(forgex_amd, forgex_amd.Program(pattern, forgex_amd.OP_SEARCH), cfg, row_len)
            except Exception as e:
                host_path = {"value": None, "error": repr(e)}
    roofline["device_copy_gbs"] = copy_gbs   # bytes read + bytes written per second of a plain device-to-device copy of the batch

    # ---- packed results + gather over RCCL (not part of `value`) -----------------------------------------------------------------
    # what a multi-GPU host does after the scan: each rank's results as ONE packed image (1 bit per flag + spans narrowed to the row
    # length, written by the search kernel itself for rows of up to 256 bytes), one gather to rank 0, unpacked there
    gather_ms = packed_step_ms = None
    gather_info = None
    unpacked = sizes = None
    if use_dist:
        packed = prog.match_device_packed(rows, spans=spans)
        for _ in range(5):
            prog.match_device_packed(rows, spans=spans, out=packed)
        barrier()
        p0 = time.perf_counter()
        for _ in range(20):
            prog.match_device_packed(rows, spans=spans, out=packed)
        torch.cuda.synchronize()
        packed_step_ms = (time.perf_counter() - p0) / 20 * 1e3
        # shards of equal size (weak scaling) travel as they are; the buffers of the gather are made once, outside the timed call
        n_gather = rows_per_gpu * world
        if args.scaling == "strong":
            n_gather = rows_all
        bufs = fxdist.gather_buffers(n_gather, row_len, spans, dev)
        fxdist.gather_packed(packed, n_gather, row_len, spans, buffers=bufs)   # warm-up (RCCL connection setup)
        barrier()
        g0 = time.perf_counter()
        res = fxdist.gather_packed(packed, n_gather, row_len, spans, buffers=bufs)
        if rank == 0:
            shards, sizes = res
            unpacked = [forgex_amd.unpack_results(img, m, row_len, spans) for img, m in zip(shards, sizes)]
        barrier()
        gather_ms = (time.perf_counter() - g0) * 1e3
        if rank == 0:
            assert sum(sizes) == n_gather and sizes[0] == rows_per_gpu
            # the gathered shard of rank 0 is what rank 0's plain call computed
            assert torch.equal(unpacked[0][0], flags) and (not spans or (torch.equal(unpacked[0][1], frm) and torch.equal(unpacked[0][2], to)))
            moved = sum(int(img.numel()) for img in shards[1:])   # bytes that crossed xGMI into the root (rank 0's own image stays on its GPU)
            gather_info = {"ms": gather_ms, "bytes_into_root": moved, "gbs_into_root": (moved / (gather_ms * 1e-3) / 1e9) if gather_ms else None,
                           "shard_rows": sizes, "bytes_per_row": (moved / max(1, sum(sizes[1:]))) if world > 1 else None,
                           "note": "one RCCL gather of every rank's packed image (1 bit per flag + spans narrowed to the row length) + unpack on the root, timed between barriers"}

    # the other ranks are done: rank 0's host-side legs (whole-batch parity, CPU baseline) need no collective
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    line = None
    if rank == 0:
        line = {
            "metric": "input GB/s scanned (.in. over 10M strings)", "value": total_bytes / dt / 1e9, "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("%s: `%s` .in. (flags%s) over %d x %d B rows per GPU, counter-based PRNG rows resident in HBM" % (
                cfg, pattern, " + (from,to) spans" if spans else " only", rows_per_gpu, row_len)) if not (world > 1 and args.scaling == "strong") else (
                "%s: `%s` .in. (flags%s) over the config's %d x %d B rows split into %d contiguous shards (rank 0: %d rows), counter-based PRNG rows resident in HBM" % (
                    cfg, pattern, " + (from,to) spans" if spans else " only", rows_all, row_len, world, rows_per_gpu)),
                "rows_per_gpu": rows_per_gpu, "rows_total": rows_all, "row_len": row_len, "pattern": pattern, "parallelism": "shard%d" % world,
                "outputs": "flag u8 + from/to int32" if spans else "flag u8", "matches_rank0": n_matches},
            "frac_of_hbm_peak": total_bytes / dt / 1e9 / (HBM_PEAK_GBS * world),
            "settled": {"value": total_bytes / dt_settled / 1e9, "ms_per_step": dt_settled / args.steps * 1e3,
                        "note": "the same %d timed steps after %d more untimed launches (clock transient over)" % (args.steps, SETTLE)},
            "roofline": roofline, "gather_ms": gather_ms, "packed_step_ms": packed_step_ms, "flags_only": flags_only, "host_path": host_path,
            "weak_scaling_extra": weak_extra,
            # multi-GPU runs prove themselves: ranks and devices as the collectives saw them, every rank's own step time, the gather's bytes
            # what happened before the W warm-up steps (outside warm-up and timed region): the program's start-up, as a service does it once
            "init": {"rows": 64, "ms": init_ms, "what": "one match call on 64 generated rows (tables uploaded, code objects loaded) + fxamd_program_reserve, "
                     "BEFORE the batch is generated; not a step of the workload (protocol since round 3: BASELINE.md section 3)"},
            "rccl_ranks": census["ranks"] if census else None, "devices": census["devices"] if census else None,
            "devices_distinct": census["distinct"] if census else None, "per_rank_ms_per_step": per_rank_ms if use_dist else None, "gather": gather_info,
        }
        threads = os.cpu_count() or 1
        if not args.no_parity:
            try:
                line["parity"] = full_parity(pattern, rows, flags, frm, to, threads)
            except Exception as e:
                line["parity"] = {"rows": 0, "mismatches": None, "checker": "failed: %r" % (e,)}
            try:   # ... and against the oracle, which shares no matching code with the product (as many rows as ~20 s of this host allow)
                line["parity"]["oracle"] = oracle_parity(pattern, rows, flags, frm, to, threads, budget_s=args.oracle_seconds)
            except Exception as e:
                line["parity"]["oracle"] = {"rows": 0, "mismatches": None, "checker": "failed: %r" % (e,)}
            # every OTHER rank's shard as it arrived on the root: the gathered + unpacked image against the host walker on that shard's
            # rows, regenerated here from (config, start, count) -- the RCCL gather and the unpack are inside what is checked
            # (FXAMD_BENCH_FORCE_DIST=1 at world size 1 checks rank 0's own gathered image the same way: the code path on a 1-GPU box)
            if unpacked is not None and (world > 1 or use_dist):
                shard_checks = []
                try:
                    del rows
                    for r in range(1 if world > 1 else 0, world):
                        if args.scaling == "strong" and not args.rows:
                            s_r = fxdist.shard_bounds(n_cfg, r, world)[0]
                        else:
                            s_r = r * rows_per_gpu
                        rows_r = synth.batch(cfg, s_r, sizes[r], dev)
                        f_r, a_r, b_r = unpacked[r] if spans else (unpacked[r][0], None, None)
                        res_r = full_parity(pattern, rows_r, f_r, a_r, b_r, threads)
                        shard_checks.append({"rank": r, "first_row": int(s_r), "rows": res_r["rows"], "mismatches": res_r["mismatches"]})
                        del rows_r
                    line["parity"]["gathered_shards"] = shard_checks
                    line["parity"]["gathered_mismatches"] = sum((c["mismatches"] or 0) for c in shard_checks)
                except Exception as e:
                    line["parity"]["gathered_shards"] = "failed: %r" % (e,)
        else:
            line["parity"] = None
        if not args.no_cpu_baseline:
            gpu_res = None
            if spans:
                k = min(rows_per_gpu, 200000)
                gpu_res = (flags[:k].cpu().numpy().astype("int64"), frm[:k].cpu().numpy().astype("int64"), to[:k].cpu().numpy().astype("int64"))
            line["cpu_baseline"] = cpu_baseline(cfg, pattern, row_len, gpu_res)
        else:
            line["cpu_baseline"] = None
    if rank == 0:
        emit(line)


if __name__ == "__main__":
    main()
