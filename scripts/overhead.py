"""Fixed per-call cost: encode/decode of a 64 KiB buffer (kernels take microseconds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raisin_amd import huffman
n = 1 << 16
g = torch.Generator(device="cuda").manual_seed(1)
src = torch.randint(0, 128, (n,), dtype=torch.uint8, device="cuda", generator=g)
out = torch.empty(n + (1 << 16), dtype=torch.uint8, device="cuda")
dec = torch.empty(n + (1 << 16), dtype=torch.uint8, device="cuda")
for _ in range(20):
    c = huffman.compress_tensor(src, out=out); d = huffman.decompress_tensor(c, out=dec)
t0 = time.perf_counter()
for _ in range(200):
    c = huffman.compress_tensor(src, out=out)
t1 = time.perf_counter()
for _ in range(200):
    d = huffman.decompress_tensor(c, out=dec)
t2 = time.perf_counter()
print("per call: encode %.1f us, decode %.1f us" % ((t1 - t0) / 200 * 1e6, (t2 - t1) / 200 * 1e6))
