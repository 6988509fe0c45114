"""Multi-GPU plumbing for the batch path: rows are independent units, so the batch shards into contiguous index
ranges with NO data-path collective; the only exchange is one gather of PACKED results to rank 0
(1 bit per flag + from/to narrowed to the row length) over RCCL -- a direct gather uses the root's 7 inbound xGMI
links in parallel.  Works with any torch.distributed backend (nccl on GPUs, gloo in the CPU tests).

On GPUs the packed image comes straight from the match call (Program.match_device_packed: the search kernel stores the
tile's ballot as the flag word and the spans narrow) and travels with gather_packed; pack_results / unpack_results here are
the same layout written with torch ops -- the CPU tests' implementation and the cross-check of the kernels' packing."""
import os
import socket

import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous [start, stop) of `rank`: shard i gets indices [i*n/world, (i+1)*n/world)."""
    return (n * rank) // world, (n * (rank + 1)) // world


def span_bytes(row_len):
    """Bytes per span entry of the packed layout -- the C ABI's rule (fxamd_packed_layout): 1 up to 255, 2 up to 65535, else 4."""
    return 1 if row_len <= 255 else (2 if row_len <= 65535 else 4)


def span_dtype(row_len):
    """Storage dtype of a span entry.  Two-byte entries are UNSIGNED 16-bit values kept in int16 storage (torch has no arithmetic
    on uint16): _narrow / _widen below convert."""
    return {1: torch.uint8, 2: torch.int16, 4: torch.int32}[span_bytes(row_len)]


def _narrow(x, row_len):
    w = span_bytes(row_len)
    if w == 2:   # 0..65535 -> the int16 with the same bit pattern
        x = x.to(torch.int32)
        return (((x + 32768) % 65536) - 32768).to(torch.int16)
    return x.to(span_dtype(row_len))


def _widen(x):
    y = x.to(torch.int32)
    return (y & 0xFFFF) if x.dtype == torch.int16 else y


def pack_results(flags, frm, to, row_len):
    """flags uint8[n] (0/1), frm/to int32[n] -> (bits uint8[ceil(n/8)], from narrow[n], to narrow[n])."""
    n = flags.shape[0]
    pad = (-n) % 8
    f = torch.cat([flags, flags.new_zeros(pad)]) if pad else flags
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=flags.device)
    bits = (f.view(-1, 8).to(torch.int32) * w).sum(dim=1).to(torch.uint8)
    return bits, _narrow(frm, row_len), _narrow(to, row_len)


def unpack_results(bits, frm, to, n):
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=bits.device)
    flags = ((bits.to(torch.int32)[:, None] & w) != 0).to(torch.uint8).reshape(-1)[:n]
    return flags, _widen(frm), _widen(to)


def packed_layout(n, row_len):
    """Byte layout of one shard's packed results: [bits | pad to 16][from, narrow][pad to 16][to, narrow] -> (off_from, off_to, total)."""
    w = span_bytes(row_len)
    nb = ((n + 7) // 8 + 15) & ~15
    ns = (n * w + 15) & ~15
    return nb, nb + ns, nb + 2 * ns


def gather_results(flags, frm, to, n_total, row_len, dst=0):
    """Every rank passes its shard's results; rank `dst` returns (flags, from, to) for all n_total rows, others None.
    ONE collective: each shard travels as a single uint8 image (bits + narrowed spans) -- byte tensors are the one dtype every
    backend moves (RCCL has no 16-bit integer type)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    bits, f8, t8 = pack_results(flags, frm, to, row_len)
    sizes = []
    for r in range(world):
        a, b = shard_bounds(n_total, r, world)
        sizes.append(b - a)
    mx = max(packed_layout(m, row_len)[2] for m in sizes)
    n = flags.shape[0]
    off_f, off_t, _ = packed_layout(n, row_len)
    buf = torch.zeros(mx, dtype=torch.uint8, device=flags.device)
    buf[:bits.numel()] = bits
    fb, tb = f8.view(torch.uint8), t8.view(torch.uint8)
    buf[off_f:off_f + fb.numel()] = fb
    buf[off_t:off_t + tb.numel()] = tb
    if rank != dst:
        dist.gather(buf, None, dst=dst)
        return None
    lst = [torch.empty(mx, dtype=torch.uint8, device=flags.device) for _ in range(world)]
    dist.gather(buf, lst, dst=dst)
    dt = span_dtype(row_len)
    w = span_bytes(row_len)
    fl, fr, tt = [], [], []
    for r in range(world):
        m = sizes[r]
        o_f, o_t, _ = packed_layout(m, row_len)
        img = lst[r]
        f, x, y = unpack_results(img[:(m + 7) // 8], img[o_f:o_f + m * w].view(dt), img[o_t:o_t + m * w].view(dt), m)
        fl.append(f)
        fr.append(x)
        tt.append(y)
    return torch.cat(fl), torch.cat(fr), torch.cat(tt)


def _shard_sizes(n_total, world):
    return [shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)]


def _packed_total(m, row_len, spans):
    w = span_bytes(row_len) if spans else 0
    nb = ((m + 7) // 8 + 15) & ~15
    return nb + 2 * ((m * w + 15) & ~15)


def gather_buffers(n_total, row_len, spans, device, dst=0):
    """What gather_packed needs besides the shard's image, allocated ONCE (outside any timed gather): the send buffer of the common
    size and, on rank `dst`, the receive buffers.  Returns (send, recv_list or None)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    mx = max(max(_packed_total(m, row_len, spans) for m in _shard_sizes(n_total, world)), 16)
    send = torch.zeros(mx, dtype=torch.uint8, device=device)
    recv = [torch.empty(mx, dtype=torch.uint8, device=device) for _ in range(world)] if rank == dst else None
    return send, recv


def gather_packed(packed, n_total, row_len, spans=True, dst=0, buffers=None):
    """Every rank passes the packed image of its shard (Program.match_device_packed); rank `dst` returns the list of the shards'
    images (views trimmed to each shard's size) together with the shards' row counts, others None.  ONE collective.
    `buffers`: gather_buffers(...) made beforehand, so that the call itself allocates nothing."""
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = _shard_sizes(n_total, world)
    mx = max(max(_packed_total(m, row_len, spans) for m in sizes), 16)
    send, recv = buffers if buffers is not None else gather_buffers(n_total, row_len, spans, packed.device, dst)
    assert send.numel() == mx and (rank != dst or len(recv) == world)
    buf = packed
    if packed.numel() != mx:   # a shard smaller than the largest one: its image rides in the common-size send buffer
        buf = send
        buf[:min(packed.numel(), mx)] = packed[:mx]
    if rank != dst:
        dist.gather(buf, None, dst=dst)
        return None
    dist.gather(buf, recv, dst=dst)
    return [recv[r][:max(_packed_total(sizes[r], row_len, spans), 16)] for r in range(world)], sizes


def device_identity(device):
    """What tells one GPU of a node from another, as a string: name, UUID and PCI address where torch exposes them, host name; for
    the CPU dry runs the process id."""
    device = torch.device(device)
    parts = []
    if device.type == "cuda":
        p = torch.cuda.get_device_properties(device)
        parts.append("name=%s" % p.name)
        for k in ("uuid", "pci_domain_id", "pci_bus_id", "pci_device_id"):
            if hasattr(p, k):
                parts.append("%s=%s" % (k, getattr(p, k)))
        parts.append("index=%d" % (device.index if device.index is not None else torch.cuda.current_device()))
    else:
        parts.append("cpu pid=%d" % os.getpid())
    parts.append("host=%s" % socket.gethostname())
    return " ".join(parts)


def job_census(device):
    """Who takes part in this job, as the COLLECTIVES see it (not as the environment claims): `ranks` = an all-reduce of 1 over the
    group, `devices` = every rank's device_identity (an all-gather of fixed-size byte tensors: byte tensors are the one dtype every
    backend moves), `distinct` = no two ranks on one device.  Every rank gets the same answer."""
    device = torch.device(device)
    one = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(one)
    me = device_identity(device).encode()[:159]
    buf = torch.zeros(160, dtype=torch.uint8, device=device)
    buf[:len(me)] = torch.tensor(list(me), dtype=torch.uint8, device=device)
    got = [torch.empty_like(buf) for _ in range(dist.get_world_size())]
    dist.all_gather(got, buf)
    devices = [bytes(g.cpu().tolist()).rstrip(b"\0").decode(errors="replace") for g in got]
    return {"ranks": int(one.item()), "devices": devices, "distinct": len(set(devices)) == len(devices), "backend": dist.get_backend()}


def gather_floats(x, device):
    """One float per rank -> the list of all ranks' values on every rank (per-rank step times of the bench)."""
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    got = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(got, t)
    return [float(g.item()) for g in got]


# ---- collective-free delivery ("7 peer writes into the root's buffer"; round 6) -------------------------------------------
# ONE process drives every GPU of the node (peer access enabled): the root's result buffer holds one packed image per shard, and each shard's
# match call is given ITS slice of that buffer as `d_packed` -- the C ABI takes any device pointer -- so the kernels of GPU i write their flag
# words and narrow spans straight into the root GPU's HBM over xGMI while they run.  No gather step, no collective, nothing to wait for after
# the scan but the stores themselves.  bench.py --single-process times it; the RCCL gather above stays the path of one-process-per-GPU hosts.
def peer_direct_layout(sizes, row_len, spans=True):
    """Offsets of the shards' packed images in the root's buffer (16-byte aligned, the C ABI's fxamd_packed_layout sizes) -> (offsets, total)."""
    offs, o = [], 0
    for m in sizes:
        offs.append(o)
        o += max(_packed_total(m, row_len, spans), 16)
    return offs, o


def pack_image(flags, frm, to, row_len, spans=True):
    """One shard's packed image as a uint8 tensor -- the layout the kernels write, with torch ops (CPU dry runs and cross-checks)."""
    n = flags.shape[0]
    bits, f8, t8 = pack_results(flags, frm if spans else flags.new_zeros(n, dtype=torch.int32), to if spans else flags.new_zeros(n, dtype=torch.int32), row_len)
    img = torch.zeros(max(_packed_total(n, row_len, spans), 16), dtype=torch.uint8, device=flags.device)
    img[:bits.numel()] = bits
    if spans:
        off_f, off_t, _ = packed_layout(n, row_len)
        fb, tb = f8.view(torch.uint8), t8.view(torch.uint8)
        img[off_f:off_f + fb.numel()] = fb
        img[off_t:off_t + tb.numel()] = tb
    return img


def unpack_image(img, n, row_len, spans=True):
    """Inverse of pack_image (torch ops): -> (flags uint8[n], from int32[n] or None, to int32[n] or None)."""
    if not spans:
        f, _, _ = unpack_results(img[:(n + 7) // 8], img.new_zeros(0), img.new_zeros(0), n)
        return f, None, None
    w, dt = span_bytes(row_len), span_dtype(row_len)
    off_f, off_t, _ = packed_layout(n, row_len)
    return unpack_results(img[:(n + 7) // 8], img[off_f:off_f + n * w].view(dt), img[off_t:off_t + n * w].view(dt), n)
