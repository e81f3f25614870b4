"""the Python-process question: does the pipelined Huffman decode overlap its transfers when its INPUT is a library
result block (2 MiB-aligned, huge pages asked for) instead of a numpy copy of it?"""
import sys, time, ctypes; sys.path.insert(0, ".")
import numpy as np
import torch, workloads as W
from raisin_amd import _lib
L = _lib.lib()
n = 1 << 30
src = W.config_input("2a", n, "cuda:0").cpu().numpy()
def raw(fn, ptr, size, *extra):
    out = ctypes.POINTER(ctypes.c_uint8)(); got = ctypes.c_size_t(0)
    t0 = time.perf_counter()
    _lib.check(fn(ptr, size, *extra, ctypes.byref(out), ctypes.byref(got)))
    return out, got.value, (time.perf_counter() - t0) * 1e3
c_blk, c_n, t = raw(L.rsn_huffman_compress, src.ctypes.data_as(ctypes.c_char_p), src.size)
print("compress %.1f ms -> %d" % (t, c_n))
c_np = np.ctypeslib.as_array(c_blk, shape=(c_n,)).copy()
import mmap
aligned = mmap.mmap(-1, (c_n + (2 << 20)) & ~((2 << 20) - 1))
try: aligned.madvise(mmap.MADV_HUGEPAGE)
except Exception as e: print("madvise:", e)
a_np = np.frombuffer(aligned, dtype=np.uint8)[:c_n]; a_np[:] = c_np
for label, ptr in (("input = the library's block", ctypes.cast(c_blk, ctypes.c_char_p)), ("input = a numpy copy", c_np.ctypes.data_as(ctypes.c_char_p)),
                   ("input = an mmap'd, huge-page-advised copy", a_np.ctypes.data_as(ctypes.c_char_p))):
    ts = []
    for _ in range(4):
        d, dn, t = raw(L.rsn_huffman_decompress, ptr, c_n)
        ts.append(round(t, 2)); L.rsn_free(d)
    print("%-45s decode ms %s" % (label, ts), flush=True)
