import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "support"))
import numpy as np, torch
import forgex_amd as fx
pat = rb"${2,}"
alpha = np.frombuffer(b"abcxyz019 .-\n\tAZ_@", dtype=np.uint8)
rng = np.random.default_rng(5)
rows = alpha[rng.integers(0, len(alpha), size=(192, 257))]
p = fx.Program(pat, fx.OP_SEARCH)
def flags_only(x):
    d = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    f, _, _ = p.match_device(d, spans=False)
    torch.cuda.synchronize()
    return f.cpu().numpy()
print("full batch row127:", flags_only(rows)[127])
print("row127 alone:", flags_only(rows[127:128]))
print("rows 64..127:", flags_only(rows[64:128])[63], " rows 127..191 first:", flags_only(rows[127:])[0])
x = rows.copy(); x[126] = rows[127]; x[127] = rows[126]
print("swapped 126<->127:", flags_only(x)[126:128])
for cut in (250, 240, 200, 130, 124, 100, 20):
    y = rows[127:128].copy(); y[0, cut:] = ord('a')
    print("row127 alone, bytes >= %d set to 'a':" % cut, flags_only(y))
for L2 in (258, 264, 265, 273, 300):
    z = np.full((1, L2), ord('a'), np.uint8); z[0, :257] = rows[127]
    print("row127 padded with 'a' to L", L2, flags_only(z))
z = rows[127:128, :256]
print("row127 cut to 256:", flags_only(np.ascontiguousarray(z)), p.last_path())
