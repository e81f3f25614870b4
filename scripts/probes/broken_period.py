"""a period broken every so often (scripts/probes/periodic_lzss.py's slow shape): where the LZSS encode's time goes"""
import sys; sys.path.insert(0, ".")
import random
import numpy as np
from raisin_amd import _lib
import bench
L = _lib.lib()
def make(n, period, seed, gap_lo=20000, gap_hi=200000):
    rng = random.Random(seed)
    alph = "abcdefghijklmnopqrstuvwxyz ,.\n"
    unit = "".join(rng.choices(alph, k=period)).encode()
    b = bytearray((unit * (n // len(unit) + 1))[:n])
    at = rng.randint(1000, 50000)
    while at < n:
        b[at] = ord(rng.choice(alph)); at += rng.randint(gap_lo, gap_hi)
    return bytes(b)
for n, period in ((1 << 20, 37), (1 << 20, 256), (1 << 23, 100)):
    data = make(n, period, 7)
    arr = np.frombuffer(data, dtype=np.uint8)
    c, _ = bench._host_call(L.rsn_lzss_compress, arr, 4096)
    sys.stderr.write("=== n %d period %d\n" % (n, period)); sys.stderr.flush()
    _lib.prof_enable(True); _lib.prof_reset()
    c, te = bench._host_call(L.rsn_lzss_compress, arr, 4096)
    pe = _lib.prof_get(); _lib.prof_enable(False)
    top = sorted(((v[1], k, v[0]) for k, v in pe.items() if v[0]), reverse=True)[:6]
    print("n %8d period %4d: %8.1f ms -> %d B, %d launches; %s" % (n, period, te, c.size, sum(v[0] for v in pe.values()), ", ".join("%s x%d %.2f ms" % (k, m, t) for t, k, m in top)), flush=True)
