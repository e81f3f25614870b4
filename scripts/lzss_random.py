"""LZSS encode of 16 MiB and 1 GiB of random bytes: the chain walk's dense hand-off to the bucket search."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from raisin_amd import lz
for mib in (16, 1024):
    d = torch.randint(0, 256, (mib << 20,), dtype=torch.uint8, device="cuda")
    c = lz.compress_tensor(d); torch.cuda.synchronize()
    t0 = time.perf_counter(); c = lz.compress_tensor(d); torch.cuda.synchronize(); print("random %d MiB: %.2f ms" % (mib, (time.perf_counter() - t0) * 1e3))
