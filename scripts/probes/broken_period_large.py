"""the slowest known shape at scale: a 37-byte period with a changed byte every 20-200 KB, device-resident LZSS encode of 64 and 256 MiB"""
import sys, time; sys.path.insert(0, ".")
import random
import numpy as np, torch
from raisin_amd import lz, _lib
rng = random.Random(7)
alph = "abcdefghijklmnopqrstuvwxyz ,.\n"
unit = "".join(rng.choices(alph, k=37)).encode()
for mib in (64, 256):
    n = mib << 20
    b = np.frombuffer((unit * (n // 37 + 1))[:n], dtype=np.uint8).copy()
    at = 30000
    while at < n:
        b[at] = ord(rng.choice(alph)); at += rng.randint(20000, 200000)
    src = torch.from_numpy(b).cuda()
    out = torch.empty(n + n // 8 + (1 << 20), dtype=torch.uint8, device="cuda")
    ts = []
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        c = lz.compress_tensor(src, out=out)
        torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t) * 1e3, 1))
    d = lz.decompress_tensor(c)
    print("%4d MiB: encode ms %s -> %d B, lossless %s" % (mib, ts, c.numel(), bool(torch.equal(d, src))), flush=True)
