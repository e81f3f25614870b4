"""the same question without torch in the process: numpy and ctypes only"""
import sys, time, ctypes; sys.path.insert(0, ".")
import numpy as np
L = ctypes.CDLL("raisin_amd/librsn.so")
for f in (L.rsn_huffman_compress, L.rsn_huffman_decompress):
    f.restype = ctypes.c_int
L.rsn_free.argtypes = [ctypes.c_void_p]
class _lib:
    @staticmethod
    def check(rc):
        assert rc == 0, rc
assert "torch" not in sys.modules
n = 1 << 30
src = np.random.default_rng(1).integers(0, 128, size=n, dtype=np.uint8)
def raw(fn, ptr, size, *extra):
    out = ctypes.POINTER(ctypes.c_uint8)(); got = ctypes.c_size_t(0)
    t0 = time.perf_counter()
    _lib.check(fn(ptr, ctypes.c_size_t(size), *extra, ctypes.byref(out), ctypes.byref(got)))
    return out, got.value, (time.perf_counter() - t0) * 1e3
c_blk, c_n, t = raw(L.rsn_huffman_compress, src.ctypes.data_as(ctypes.c_char_p), src.size)
ts = []
for _ in range(5):
    d, dn, t = raw(L.rsn_huffman_decompress, ctypes.cast(c_blk, ctypes.c_char_p), c_n)
    ts.append(round(t, 2)); L.rsn_free(d)
print("no torch in the process (%s): decode ms %s" % ("torch" in sys.modules, ts))
import threading
res = []
def work():
    for _ in range(4):
        d, dn, t = raw(L.rsn_huffman_decompress, ctypes.cast(c_blk, ctypes.c_char_p), c_n)
        res.append(round(t, 2)); L.rsn_free(d)
th = threading.Thread(target=work); th.start(); th.join()
print("from a second Python thread: decode ms", res)
