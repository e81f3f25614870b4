"""host-buffer calls over sizes and kinds of data: median of 5 warm calls each, to find the inputs whose time is out of line with
their neighbours' (us).  huffman encode / decode, lzss encode / decode."""
import sys; sys.path.insert(0, ".")
import random
import numpy as np
from raisin_amd import _lib
import bench
L = _lib.lib()
sam = open("tests/golden/samiam.txt", "rb").read()
rng = random.Random(5)
words = ["".join(rng.choice("etaoinshrdlucmfwypvbgkqjxz") for _ in range(rng.randint(1, 9))) for _ in range(3000)]
def text(n):
    out, size = [], 0
    while size < n:
        w = rng.choice(words) + " "; out.append(w); size += len(w)
    return "".join(out).encode()[:n]
kinds = {
    "text": text,
    "samiam x": lambda n: (sam * (n // len(sam) + 1))[:n],
    "period 7": lambda n: (b"abcabda" * (n // 7 + 1))[:n],
    "period 5000": lambda n: (text(5000) * (n // 5000 + 1))[:n],
    "zeros": lambda n: bytes(n),
    "random ascii": lambda n: np.random.default_rng(n).integers(0, 128, size=n, dtype=np.uint8).tobytes(),
    "random bytes": lambda n: np.random.default_rng(n + 1).integers(0, 256, size=n, dtype=np.uint8).tobytes(),
    "utf8": lambda n: ("héllo wörld 世界 " * (n // 20 + 1)).encode()[:n],
}
sizes = [1 << 10, 1 << 13, 1 << 16, 3 << 16, 1 << 20, 1 << 22, 1 << 24]
def med(fn, *a):
    ts = []
    r = None
    for _ in range(5):
        r, t = bench._host_call(fn, *a)
        ts.append(t)
    return r, sorted(ts)[2] * 1e3
print("%-14s %9s | %9s %9s | %9s %9s" % ("kind", "bytes", "huff enc", "huff dec", "lzss enc", "lzss dec"))
for name, gen in kinds.items():
    for n in sizes:
        data = gen(n)
        arr = np.frombuffer(data, dtype=np.uint8)
        try:
            c, he = med(L.rsn_huffman_compress, arr)
            d, hd = med(L.rsn_huffman_decompress, c)
        except Exception as e:
            he = hd = float("nan")
        try:
            c, le = med(L.rsn_lzss_compress, arr, 4096)
            d, ld = med(L.rsn_lzss_decompress, c)
            ok = d.tobytes() == data
        except Exception as e:
            le = ld = float("nan"); ok = False
        print("%-14s %9d | %9.0f %9.0f | %9.0f %9.0f %s" % (name, n, he, hd, le, ld, "" if ok else "LZSS MISMATCH"), flush=True)
