"""Huffman encode with the small tiles (set SMALL_INPUT in huff_encode.hip to 64 MiB to see where they stop paying) against RSN_HUFF_NO_SMALL_TILES=1: run twice."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from raisin_amd import huffman
from test_gpu_lzss import text
for n in (4096, 65536, 1 << 20, 8 << 20, 64 << 20, 65 << 20):
    for name, d in (("utf-8 text", torch.frombuffer(bytearray(text(5, min(n, 8 << 20)) * max(1, n >> 23)), dtype=torch.uint8).cuda()),
                    ("ascii", torch.randint(32, 127, (n,), dtype=torch.uint8, device="cuda"))):
        c = huffman.compress_tensor(d); torch.cuda.synchronize()
        reps = 30
        t0 = time.perf_counter()
        for _ in range(reps): c = huffman.compress_tensor(d)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print("%-10s %9d B: encode %8.1f us" % (name, d.numel(), (t1 - t0) / reps * 1e6))
