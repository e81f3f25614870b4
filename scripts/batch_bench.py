"""rsn_huffman_compress_batch on 8 x 256 MiB host buffers: the upload / encode / download pipeline, and the same chunks as a loop of single calls."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import workloads as W
from raisin_amd import _lib

L = _lib.lib()
n, k = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20, 8
bufs = [W.config_input("5", n, "cuda", chunk=i).cpu().numpy() for i in range(k)]   # generated on the device: the CPU generator takes a minute
ins = (ctypes.c_char_p * k)(*[ctypes.cast(b.ctypes.data, ctypes.c_char_p) for b in bufs])
lens = (ctypes.c_size_t * k)(*[n] * k)
outs = (ctypes.POINTER(ctypes.c_uint8) * k)()
olens = (ctypes.c_size_t * k)()
for rep in range(3):
    t0 = time.perf_counter()
    _lib.check(L.rsn_huffman_compress_batch(k, ins, lens, outs, olens))
    t = time.perf_counter() - t0
    for i in range(k):
        L.rsn_free(outs[i])
    print("pipelined batch of %d x %d MiB: %.1f ms (%.2f GB/s)" % (k, n >> 20, t * 1e3, k * n / t / 1e9))
one, one_n = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(k):
        _lib.check(L.rsn_huffman_compress(ins[i], n, ctypes.byref(one), ctypes.byref(one_n)))
        L.rsn_free(one)
    t = time.perf_counter() - t0
    print("loop of single calls, %d x %d MiB: %.1f ms (%.2f GB/s)" % (k, n >> 20, t * 1e3, k * n / t / 1e9))
