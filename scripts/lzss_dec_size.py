"""LZSS decode of config 4's text at one size (MiB), three timed calls and the kernels of a fourth."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import workloads as W
from raisin_amd import lz, _lib
n = int(sys.argv[1]) << 20
d = W.config_input("4", n, "cuda")
c = lz.compress_tensor(d); o = lz.decompress_tensor(c); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); o = lz.decompress_tensor(c); torch.cuda.synchronize(); print("decode %d MiB: %.2f ms" % (n >> 20, (time.perf_counter() - t0) * 1e3))
_lib.prof_enable(True); _lib.prof_reset()
o = lz.decompress_tensor(c); torch.cuda.synchronize()
for k, (cnt, ms) in sorted(_lib.prof_get().items()): print("      %-24s %2d  %9.1f us" % (k, cnt, ms * 1e3))
