"""Where a small call's time goes: kernel launches and their device time against the call's wall time (text, both codecs)."""
import os
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch

from raisin_amd import _lib, huffman, lz
from test_gpu_lzss import text

sizes = [int(a) for a in sys.argv[1:]] or [65536, 1 << 20]
for n in sizes:
    d = torch.frombuffer(bytearray(text(5, n)), dtype=torch.uint8).cuda()
    for mod, name in ((lz, "lzss"), (huffman, "huffman")):
        c = mod.compress_tensor(d); o = mod.decompress_tensor(c); torch.cuda.synchronize()
        for what, fn, arg in (("encode", mod.compress_tensor, d), ("decode", mod.decompress_tensor, c)):
            _lib.prof_enable(False)
            t0 = time.perf_counter()
            for _ in range(20): fn(arg)
            torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20
            _lib.prof_enable(True); _lib.prof_reset()
            fn(arg); torch.cuda.synchronize()
            p = _lib.prof_get()
            print("%-8s %s %8d B: wall %7.1f us, %2d launches, %7.1f us on the device" % (name, what, n, wall * 1e6, sum(v[0] for v in p.values()), sum(v[1] for v in p.values()) * 1e3))
            for k, (cnt, ms) in sorted(p.items()): print("      %-24s %2d  %7.1f us" % (k, cnt, ms * 1e3))
_lib.prof_enable(False)
