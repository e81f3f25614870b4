"""LZSS encode of 64 MiB of runs of equal bytes (four letters, run length from argv, default 37), with its kernels."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from raisin_amd import lz, _lib
n = 64 << 20
g = torch.Generator(device="cuda"); g.manual_seed(5)
runlen = int(sys.argv[1]) if len(sys.argv) > 1 else 37
d = torch.repeat_interleave(torch.randint(97, 101, (n // runlen + 1,), device="cuda", generator=g, dtype=torch.uint8), runlen)[:n].contiguous()
c = lz.compress_tensor(d); torch.cuda.synchronize()
_lib.prof_enable(True); _lib.prof_reset()
t0 = time.perf_counter(); c = lz.compress_tensor(d); torch.cuda.synchronize(); print("runs of %d, 64 MiB: %.2f ms" % (runlen, (time.perf_counter() - t0) * 1e3))
for k, (cnt, ms) in sorted(_lib.prof_get().items()): print("      %-24s %2d  %9.1f us" % (k, cnt, ms * 1e3))
