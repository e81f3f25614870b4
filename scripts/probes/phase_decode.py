"""the inputs whose Huffman decode did not settle (scripts/probes/periodic_decode.py found them): times, and that the bytes are right"""
import sys; sys.path.insert(0, ".")
import numpy as np
from raisin_amd import _lib
from oracle import oracle as O
import bench
L = _lib.lib()
unit_utf8 = b'a\xe4\xb8\x96\xc3\xa8\xe6\x9c\xac\xe4\xb8\x96u\xc3\xb6uuu\xe6\x9c\xac\xc3\xa8\xe6\x9c\xac'
cases = {"utf8 unit of 13 runes, 4 MiB": (unit_utf8 * ((4 << 20) // len(unit_utf8) + 1))[:4 << 20],
         "utf8 'héllo wörld 世界 ', 4 MiB": ("héllo wörld 世界 " * (4 << 18)).encode()[:4 << 20],
         "utf8 'héllo wörld 世界 ', 64 KiB": ("héllo wörld 世界 " * 5000).encode()[:65536],
         "even lengths (2 phases), 2 MiB": None, "lengths 5 and 10 (5 phases), 2 MiB": None}
rng = np.random.default_rng(3)
# four symbols of weight 4, sixteen of weight 1 -> lengths 2 ... hmm: build by explicit frequencies below
def by_freq(freqs, n, seed):
    syms = np.repeat(np.arange(len(freqs)) + 48, freqs)
    r = np.random.default_rng(seed)
    out = r.choice(syms, size=n)
    return out.astype(np.uint8).tobytes()
cases["even lengths (2 phases), 2 MiB"] = by_freq([16] * 3 + [4] * 4, 2 << 20, 1)      # lengths 2,2,2,4,4,4,4
cases["lengths 5 and 10 (5 phases), 2 MiB"] = by_freq([32] * 31 + [1] * 32, 2 << 20, 2)
for name, data in cases.items():
    arr = np.frombuffer(data, dtype=np.uint8)
    c, _ = bench._host_call(L.rsn_huffman_compress, arr)
    ts = []
    for _ in range(3):
        d, t = bench._host_call(L.rsn_huffman_decompress, c)
        ts.append(t)
    want = O.huffman_decompress(c.tobytes())
    print("%-40s decode %9.2f ms (first %9.2f)  == oracle: %s" % (name, sorted(ts)[1], ts[0], d.tobytes() == want), flush=True)
