import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from raisin_amd import _lib
print("torch loaded:", "torch" in sys.modules)
L = _lib.lib()
for f in (L.rsn_huffman_compress, L.rsn_huffman_decompress): f.argtypes = None
def call(fn, buf):
    out = ctypes.POINTER(ctypes.c_uint8)(); n = ctypes.c_size_t(0)
    t0 = time.perf_counter()
    _lib.check(fn(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), buf.size, ctypes.byref(out), ctypes.byref(n)))
    t1 = time.perf_counter()
    res = np.ctypeslib.as_array(out, shape=(n.value,)).copy()
    L.rsn_free(out)
    return res, (t1 - t0) * 1e3
src = np.random.default_rng(1).integers(0, 128, size=1 << 30, dtype=np.uint8)
c, _ = call(L.rsn_huffman_compress, src)
ts = []
for rep in range(4):
    d, t = call(L.rsn_huffman_decompress, c); ts.append(t)
print("2a-like: huffman decompress of 1 GiB host->host: %s ms, lossless=%s" % (" ".join("%.1f" % t for t in ts), bool(np.array_equal(d, src))))
if len(sys.argv) > 1:
    import torch
    print("torch imported; cuda available:", torch.cuda.is_available())
    ts = []
    for rep in range(4):
        d, t = call(L.rsn_huffman_decompress, c); ts.append(t)
    print("after importing torch: %s ms" % " ".join("%.1f" % t for t in ts))

# the same call on the LIBRARY's block (the compressor's result, not a numpy copy of it) and on a numpy copy made without huge pages
out = ctypes.POINTER(ctypes.c_uint8)(); n = ctypes.c_size_t(0)
_lib.check(L.rsn_huffman_compress(src.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), src.size, ctypes.byref(out), ctypes.byref(n)))
ts = []
for rep in range(4):
    o2 = ctypes.POINTER(ctypes.c_uint8)(); n2 = ctypes.c_size_t(0)
    t0 = time.perf_counter()
    _lib.check(L.rsn_huffman_decompress(out, n.value, ctypes.byref(o2), ctypes.byref(n2)))
    ts.append((time.perf_counter() - t0) * 1e3)
    L.rsn_free(o2)
print("input = the library's own result block: %s ms" % " ".join("%.1f" % t for t in ts))
import mmap
mm = mmap.mmap(-1, n.value + 4096)
buf = np.frombuffer(mm, dtype=np.uint8, count=n.value)
buf[:] = np.ctypeslib.as_array(out, shape=(n.value,))
ts = []
for rep in range(4):
    d, t = call(L.rsn_huffman_decompress, buf); ts.append(t)
print("input = an anonymous mmap (no madvise): %s ms" % " ".join("%.1f" % t for t in ts))
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
