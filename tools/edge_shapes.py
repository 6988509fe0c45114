import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "support"))
import numpy as np
import forgex_amd as fx
import oracle_lib
NT = 8
for pat in (rb"a*", rb"[a-z]+\d+", rb" ", rb"b*", rb"^$", rb"."):
    for n, L in ((4, 0), (6, 1), (3, 2)):
        rows = np.zeros((n, L), dtype=np.uint8)
        if L:
            rows[:] = np.frombuffer((b" ab1" * 4)[:L], dtype=np.uint8)
            rows[0, :] = ord("a")
        for op, oop in ((fx.OP_SEARCH, 2), (fx.OP_MATCH, 1)):
            p = fx.Program(pat, op)
            f, a, b = p.match_host(rows, spans=(op == fx.OP_SEARCH))
            of, oa, ob = oracle_lib.batch(oop, pat, rows, NT)
            ok = np.array_equal(f, of) and (op != fx.OP_SEARCH or (np.array_equal(a, oa) and np.array_equal(b, ob)))
            print(pat, n, L, "search" if op == fx.OP_SEARCH else "match", "path", p.last_path(), "OK" if ok else ("DIFF", f, of, a, oa, b, ob))
