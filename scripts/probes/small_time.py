import sys; sys.path.insert(0, ".")
import numpy as np, time
from raisin_amd import _lib
import bench
L = _lib.lib()
data = open("tests/golden/samiam.txt","rb").read()
data = (data * (65536 // len(data) + 1))[:65536]
arr = np.frombuffer(data, dtype=np.uint8)
for rep in range(2):
    enc, dec = [], []
    for _ in range(200):
        c, te = bench._host_call(L.rsn_huffman_compress, arr)
        d, td = bench._host_call(L.rsn_huffman_decompress, c)
        enc.append(te); dec.append(td)
    print("64 KiB: encode median %.1f us min %.1f, decode median %.1f us min %.1f" % (np.median(enc)*1e3, min(enc)*1e3, np.median(dec)*1e3, min(dec)*1e3), d.tobytes() == data)
_lib.prof_enable(True); _lib.prof_reset()
for _ in range(50):
    c, te = bench._host_call(L.rsn_huffman_compress, arr)
    d, td = bench._host_call(L.rsn_huffman_decompress, c)
for k, v in _lib.prof_get().items():
    if v[0]: print(k, v[0], "launches, %.1f us each" % (v[1] / v[0] * 1e3))
_lib.prof_enable(False)
