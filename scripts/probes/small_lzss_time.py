"""host-buffer LZSS calls on 64 KiB and 1 MiB of the README's text: median of 50, and the launches of one call"""
import sys; sys.path.insert(0, ".")
import numpy as np
from raisin_amd import _lib
import bench
L = _lib.lib()
sam = open("tests/golden/samiam.txt", "rb").read()
for size in (65536, 1 << 20):
    data = (sam * (size // len(sam) + 1))[:size]
    arr = np.frombuffer(data, dtype=np.uint8)
    enc, dec = [], []
    for _ in range(50):
        c, te = bench._host_call(L.rsn_lzss_compress, arr, 4096)
        d, td = bench._host_call(L.rsn_lzss_decompress, c)
        enc.append(te); dec.append(td)
    _lib.prof_enable(True); _lib.prof_reset()
    c, te = bench._host_call(L.rsn_lzss_compress, arr, 4096)
    pe = _lib.prof_get(); _lib.prof_reset()
    d, td = bench._host_call(L.rsn_lzss_decompress, c)
    pd = _lib.prof_get(); _lib.prof_enable(False)
    print(size, "lzss encode median %.1f us, decode %.1f us" % (np.median(enc) * 1e3, np.median(dec) * 1e3), d.tobytes() == data, c.size)
    print("  encode launches:", {k: v[0] for k, v in pe.items() if v[0]})
    print("  decode launches:", {k: v[0] for k, v in pd.items() if v[0]})
