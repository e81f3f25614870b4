import sys, os, random
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from raisin_amd import lz
from oracle import oracle
from test_gpu_lzss import text, long_copies
import workloads as W
bad = 0
for seed in range(24):
    rng = random.Random(seed)
    parts = []
    for _ in range(rng.randint(3, 9)):
        kind = rng.choice("tttzzcnp")
        n = rng.choice((5000, 70000, 300000, 900000))
        if kind == "t": parts.append(text(rng.randrange(1 << 30), n))
        elif kind == "z": parts.append(bytes(W.zipf_text(n, seed=rng.randrange(1 << 30)).numpy()))
        elif kind == "c": parts.append(long_copies(rng.randrange(1000), n))
        elif kind == "n": parts.append(np.random.default_rng(seed).integers(0, 256, size=min(n, 90000), dtype=np.uint8).tobytes())
        else:
            per = rng.choice((7, 200, 4096, 5000)); blk = bytes(rng.randrange(97, 123) for _ in range(per)); parts.append((blk * (n // per + 1))[:n])
    data = b"".join(parts)
    w = rng.choice((4096, 4096, 4096, 300, 1024))
    c = lz.CompressAsync(data, False, w)
    want = oracle.lzss_compress_mt(data, w, oracle.host_cores(), 4096)
    ok = c == want and lz.Decompress(c) == data
    print(seed, len(data), w, "ok" if ok else "MISMATCH", flush=True)
    bad += not ok
print("bad:", bad)
sys.exit(1 if bad else 0)
