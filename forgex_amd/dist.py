"""Multi-GPU plumbing for the batch path: rows are independent units, so the batch shards into contiguous index
ranges with NO data-path collective; the only exchange is one gather of PACKED results to rank 0
(1 bit per flag + from/to narrowed to the row length) over RCCL -- a direct gather uses the root's 7 inbound xGMI
links in parallel.  Works with any torch.distributed backend (nccl on GPUs, gloo in the CPU tests).

On GPUs the packed image comes straight from the match call (Program.match_device_packed: the search kernel stores the
tile's ballot as the flag word and the spans narrow) and travels with gather_packed; pack_results / unpack_results here are
the same layout written with torch ops -- the CPU tests' implementation and the cross-check of the kernels' packing."""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous [start, stop) of `rank`: shard i gets indices [i*n/world, (i+1)*n/world)."""
    return (n * rank) // world, (n * (rank + 1)) // world


def span_dtype(row_len):
    return torch.uint8 if row_len <= 255 else (torch.int16 if row_len <= 32767 else torch.int32)


def pack_results(flags, frm, to, row_len):
    """flags uint8[n] (0/1), frm/to int32[n] -> (bits uint8[ceil(n/8)], from narrow[n], to narrow[n])."""
    n = flags.shape[0]
    pad = (-n) % 8
    f = torch.cat([flags, flags.new_zeros(pad)]) if pad else flags
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=flags.device)
    bits = (f.view(-1, 8).to(torch.int32) * w).sum(dim=1).to(torch.uint8)
    dt = span_dtype(row_len)
    return bits, frm.to(dt), to.to(dt)


def unpack_results(bits, frm, to, n):
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=bits.device)
    flags = ((bits.to(torch.int32)[:, None] & w) != 0).to(torch.uint8).reshape(-1)[:n]
    return flags, frm.to(torch.int32), to.to(torch.int32)


def packed_layout(n, row_len):
    """Byte layout of one shard's packed results: [bits | pad to 16][from, narrow][pad to 16][to, narrow] -> (off_from, off_to, total)."""
    w = torch.empty(0, dtype=span_dtype(row_len)).element_size()
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
    w = torch.empty(0, dtype=dt).element_size()
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


def gather_packed(packed, n_total, row_len, spans=True, dst=0):
    """Every rank passes the packed image of its shard (Program.match_device_packed); rank `dst` returns the list of the shards'
    images (views trimmed to each shard's size) together with the shards' row counts, others None.  ONE collective."""
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = []
    for r in range(world):
        a, b = shard_bounds(n_total, r, world)
        sizes.append(b - a)
    w = torch.empty(0, dtype=span_dtype(row_len)).element_size() if spans else 0

    def total(m):
        nb = ((m + 7) // 8 + 15) & ~15
        return nb + 2 * ((m * w + 15) & ~15)
    mx = max(max(total(m) for m in sizes), 16)
    buf = packed
    if packed.numel() != mx:
        buf = torch.zeros(mx, dtype=torch.uint8, device=packed.device)
        buf[:min(packed.numel(), mx)] = packed[:mx]
    if rank != dst:
        dist.gather(buf, None, dst=dst)
        return None
    lst = [torch.empty(mx, dtype=torch.uint8, device=packed.device) for _ in range(world)]
    dist.gather(buf, lst, dst=dst)
    return [lst[r][:max(total(sizes[r]), 16)] for r in range(world)], sizes
