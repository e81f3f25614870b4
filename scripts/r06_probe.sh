cd $GRAFT_REPO_ROOT
timeout 900 python - <<'PY'
import sys, time, random
sys.path.insert(0, '.')
from raisin_amd import _lib, huffman
from oracle import oracle as O
O.build()
rng = random.Random(5)
bad = 0
for n in list(range(1, 130)) + [200, 333, 1000]:
    for alph in (b"ab", b"abcdefgh \n", bytes(range(32, 127))):
        d = bytes(rng.choice(alph) for _ in range(n))
        try:
            c = huffman.Compress(d)
        except Exception as e:
            try: O.huffman_compress(d); bad += 1; print("lib error, oracle ok", n, e)
            except Exception: pass
            continue
        if c != O.huffman_compress(d): bad += 1; print("DIFF", n, alph[:4])
        if huffman.Decompress(c) != O.huffman_decompress(c): bad += 1; print("DEC DIFF", n)
print("bad", bad)
for d in (b"Hello world!\n", b"abcabcabcabcabcabcabcabc\n"):
    c = huffman.Compress(d)
    te, td = [], []
    for _ in range(30):
        t0 = time.perf_counter(); huffman.Compress(d); te.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter(); huffman.Decompress(c); td.append((time.perf_counter() - t0) * 1e3)
    _lib.prof_enable(True); _lib.prof_reset(); huffman.Compress(d); pe = _lib.prof_get(); _lib.prof_reset(); huffman.Decompress(c); pd = _lib.prof_get(); _lib.prof_enable(False)
    print(len(d), "encode ms", round(sorted(te)[15], 4), list(pe), "decode ms", round(sorted(td)[15], 4), list(pd))
PY
