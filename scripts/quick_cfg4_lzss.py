import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, workloads as W
from raisin_amd import _lib, lz
n = 1024 << 20
src = W.config_input("4", n, "cuda")
c = lz.compress_tensor(src)
torch.cuda.synchronize()
_lib.prof_enable(True); _lib.prof_reset()
t0 = time.perf_counter(); c = lz.compress_tensor(src); torch.cuda.synchronize(); t1 = time.perf_counter()
print("config 4 lzss layer: %.1f ms -> %d B" % ((t1 - t0) * 1e3, c.numel()))
for k, (cnt, ms) in sorted(_lib.prof_get().items()):
    print("  %-22s %3d launches  %.3f ms" % (k, cnt, ms))
