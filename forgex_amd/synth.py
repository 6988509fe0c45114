"""Synthetic row generators for the BASELINE.json configs.  Counter-based: every byte is a
pure function of (seed, row index, byte index) through splitmix64, so the host (oracle sample) and every GPU
shard regenerate identical rows without transfers.  Written with torch ops that behave identically on CPU and
CUDA tensors (int64 wrap-around arithmetic, logical shifts emulated by masking)."""
import torch

_C1 = 0x9E3779B97F4A7C15 - (1 << 64)
_C2 = 0xBF58476D1CE4E5B9 - (1 << 64)
_C3 = 0x94D049BB133111EB - (1 << 64)


def _lsr(z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)


def _mix(z):
    z = z + _C1
    z = (z ^ _lsr(z, 30)) * _C2
    z = (z ^ _lsr(z, 27)) * _C3
    return z ^ _lsr(z, 31)


def _cells(idx, L, seed):
    i = idx.to(torch.int64)[:, None]
    j = torch.arange(L, dtype=torch.int64, device=idx.device)[None, :]
    return _mix(i * 4096 + j + seed * 0x632BE59BD9B4E019 % (1 << 62))


def _rowhash(idx, seed, salt):
    return _mix(idx.to(torch.int64) * 4096 + 4000 + salt + seed * 0x632BE59BD9B4E019 % (1 << 62))


PATTERNS = {
    "cfg1": r"\d{3}-\d{4}",
    "cfg2": r"foo(bar|baz)",
    "cfg3": r"[a-z]+\d+",
    "cfg4": "[α-ωぁ-ん]+",
    "cfg5": r"[a-z]+\d+",
}
SHAPES = {"cfg1": (1000, 8), "cfg2": (1 << 20, 64), "cfg3": (10_000_000, 256), "cfg4": (1 << 20, 192), "cfg5": (100_000_000, 128)}
SEEDS = {"cfg1": 1, "cfg2": 2, "cfg3": 3, "cfg4": 4, "cfg5": 5}


def _letters_digits(idx, L, seed, lo, hi):
    """cfg3 / cfg5: a-z (p=.9) / blank (p=.1); half of the rows get 1-3 digits planted at offset lo..hi with a letter
    forced in front (a late, guaranteed match); the other rows hold no digit (full-scan no-match)."""
    h = _cells(idx, L, seed)
    blank = (_lsr(h, 11) % 10) == 0
    out = torch.where(blank, torch.full_like(h, 32), 97 + (_lsr(h, 40) % 26))
    r = _rowhash(idx, seed, 0)
    is_match = (r & 1) == 1
    nd = 1 + (_lsr(r, 1) % 3)
    off = lo + (_lsr(r, 8) % (hi - lo + 1))
    j = torch.arange(L, dtype=torch.int64, device=idx.device)[None, :]
    dig = is_match[:, None] & (j >= off[:, None]) & (j < (off + nd)[:, None])
    out = torch.where(dig, 48 + (_lsr(h, 20) % 10), out)
    pre = is_match[:, None] & (j == (off - 1)[:, None])
    out = torch.where(pre, 97 + (_lsr(h, 40) % 26), out)
    return out.to(torch.uint8)


def rows(cfg, idx):
    """Rows `idx` (1-D integer tensor, any device) of config `cfg` -> uint8 tensor [len(idx), L]."""
    n_total, L = SHAPES[cfg]
    seed = SEEDS[cfg]
    dev = idx.device
    if cfg == "cfg1":   # digits, '-' at byte 3 on even rows
        h = _cells(idx, L, seed)
        out = 48 + (_lsr(h, 13) % 10)
        j = torch.arange(L, dtype=torch.int64, device=dev)[None, :]
        dash = ((idx.to(torch.int64) % 2) == 0)[:, None] & (j == 3)
        return torch.where(dash, torch.full_like(out, 45), out).to(torch.uint8)
    if cfg == "cfg2":   # a-z; p=.1 plant foobar/foobaz at offset 0..58
        h = _cells(idx, L, seed)
        out = 97 + (_lsr(h, 17) % 26)
        r = _rowhash(idx, seed, 0)
        plant = (_lsr(r, 3) % 10) == 0
        off = _lsr(r, 16) % 59
        which = (_lsr(r, 40) & 1)
        j = torch.arange(L, dtype=torch.int64, device=dev)[None, :]
        rel = j - off[:, None]
        word_a = torch.tensor(list(b"foobar"), dtype=torch.int64, device=dev)
        word_b = torch.tensor(list(b"foobaz"), dtype=torch.int64, device=dev)
        inside = plant[:, None] & (rel >= 0) & (rel < 6)
        relc = rel.clamp(0, 5)
        w = torch.where(which[:, None] == 1, word_b[relc], word_a[relc])
        return torch.where(inside, w, out).to(torch.uint8)
    if cfg == "cfg3":
        return _letters_digits(idx, L, seed, 192, 252)
    if cfg == "cfg5":
        return _letters_digits(idx, L, seed, 96, 124)
    if cfg == "cfg4":   # alternating 2-byte alpha..omega and 3-byte hiragana; 10% ASCII-only rows; 1% one corrupt byte
        slots = 38
        i = idx.to(torch.int64)[:, None]
        s = torch.arange(slots, dtype=torch.int64, device=dev)[None, :]
        h = _mix(i * 4096 + s + seed * 0x632BE59BD9B4E019 % (1 << 62))
        cp2 = 0x3B1 + (_lsr(h, 9) % 25)
        cp3 = 0x3041 + (_lsr(h, 33) % 83)
        b = torch.stack([0xC0 | (cp2 >> 6), 0x80 | (cp2 & 0x3F), 0xE0 | (cp3 >> 12), 0x80 | ((cp3 >> 6) & 0x3F), 0x80 | (cp3 & 0x3F)], dim=2)
        body = b.reshape(idx.shape[0], slots * 5)
        out = torch.cat([body, torch.full((idx.shape[0], L - slots * 5), 32, dtype=torch.int64, device=dev)], dim=1)
        r = _rowhash(idx, seed, 0)
        ascii_row = (_lsr(r, 5) % 10) == 0
        ha = _cells(idx, L, seed + 100)
        out = torch.where(ascii_row[:, None], 97 + (_lsr(ha, 17) % 26), out)
        corrupt = (_lsr(r, 20) % 100) == 0
        pos = _lsr(r, 30) % L
        j = torch.arange(L, dtype=torch.int64, device=dev)[None, :]
        bad = corrupt[:, None] & (j == pos[:, None])
        out = torch.where(bad, 0x80 + (_lsr(r, 44) % 0x80)[:, None].expand_as(out), out)
        return out.to(torch.uint8)
    raise KeyError(cfg)


def batch(cfg, start, count, device, chunk=1 << 20):
    """Rows [start, start+count) of `cfg` as one contiguous uint8 tensor on `device`, generated in chunks."""
    _, L = SHAPES[cfg]
    out = torch.empty((count, L), dtype=torch.uint8, device=device)
    for c0 in range(0, count, chunk):
        c1 = min(count, c0 + chunk)
        out[c0:c1] = rows(cfg, torch.arange(start + c0, start + c1, dtype=torch.int64, device=device))
    return out
