"""Throughput of the HOST-buffer C ABI (what the cgo shim calls): pageable buffer in, malloc'ed
buffer out, PCIe both ways.  Never what bench.py reports; DESIGN.md section 5 quotes it."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from raisin_amd import _lib


def call(fn, buf, *extra):
    L = _lib.lib()
    out = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_size_t(0)
    t0 = time.perf_counter()
    _lib.check(fn(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), buf.size, *extra, ctypes.byref(out), ctypes.byref(n)))
    t1 = time.perf_counter()
    res = np.ctypeslib.as_array(out, shape=(n.value,)).copy()      # what C.GoBytes does
    t2 = time.perf_counter()
    L.rsn_free(out)
    return res, (t1 - t0) * 1e3, (t2 - t1) * 1e3


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    L = _lib.lib()
    for f in (L.rsn_huffman_compress, L.rsn_huffman_decompress, L.rsn_lzss_compress, L.rsn_lzss_decompress):
        f.argtypes = None
    rng = np.random.default_rng(2)
    src = (rng.integers(0, 128, size=mib << 20, dtype=np.uint8))
    for rep in range(3):
        c, t_c, t_cc = call(L.rsn_huffman_compress, src)
        d, t_d, t_dc = call(L.rsn_huffman_decompress, c)
        print("huffman %d MiB host->host: compress %.1f ms (%.2f GB/s), decompress %.1f ms (%.2f GB/s); caller-side copy %.1f / %.1f ms; lossless=%s"
              % (mib, t_c, src.size / t_c / 1e6, t_d, src.size / t_d / 1e6, t_cc, t_dc, bool(np.array_equal(d, src))))


if __name__ == "__main__":
    main()
