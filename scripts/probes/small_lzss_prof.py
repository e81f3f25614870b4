"""where a host-buffer LZSS encode of a small input spends its time: kernel totals of one call, for several 64 KiB inputs"""
import sys; sys.path.insert(0, ".")
import numpy as np, random
from raisin_amd import _lib
import bench
L = _lib.lib()
sam = open("tests/golden/samiam.txt", "rb").read()
rng = random.Random(5)
words = ["".join(rng.choice("etaoinshrdlucmfwypvbgkqjxz") for _ in range(rng.randint(1, 9))) for _ in range(2000)]
text = " ".join(rng.choice(words) for _ in range(20000)).encode()[:65536]
cases = {"samiam.txt alone (%d B)" % len(sam): sam, "samiam repeated to 64 KiB": (sam * 100)[:65536], "random words 64 KiB": text,
         "samiam repeated to 16 KiB": (sam * 100)[:16384], "zeros 64 KiB": bytes(65536), "period 7, 64 KiB": (b"abcabda" * 10000)[:65536]}
for name, data in cases.items():
    arr = np.frombuffer(data, dtype=np.uint8)
    for _ in range(3):
        c, te = bench._host_call(L.rsn_lzss_compress, arr, 4096)
    ts = sorted(bench._host_call(L.rsn_lzss_compress, arr, 4096)[1] for _ in range(9))
    _lib.prof_enable(True); _lib.prof_reset()
    c, te = bench._host_call(L.rsn_lzss_compress, arr, 4096)
    pe = _lib.prof_get(); _lib.prof_enable(False)
    top = sorted(((v[1], k, v[0]) for k, v in pe.items() if v[0]), reverse=True)[:5]
    print("%-30s %8.1f us  -> %6d B, %3d launches; top: %s" % (name, ts[4] * 1e3, c.size, sum(v[0] for v in pe.values()), ", ".join("%s x%d %.0f us" % (k, m, t * 1e3) for t, k, m in top)))
