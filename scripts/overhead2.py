"""Wall time per call with and without the per-kernel event records (rsn_prof)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raisin_amd import _lib, huffman
n = 1 << 30
g = torch.Generator(device="cuda").manual_seed(0x5EED0002)
src = torch.randint(0, 128, (n,), dtype=torch.uint8, device="cuda", generator=g)
out = torch.empty(n + n // 4 + (1 << 20), dtype=torch.uint8, device="cuda")
dec = torch.empty(n + (1 << 20), dtype=torch.uint8, device="cuda")
c = huffman.compress_tensor(src, out=out)
for prof in (False, True, False, True):
    _lib.prof_enable(prof); _lib.prof_reset()
    for _ in range(3): c = huffman.compress_tensor(src, out=out); d = huffman.decompress_tensor(c, out=dec)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): c = huffman.compress_tensor(src, out=out)
    t1 = time.perf_counter()
    for _ in range(20): d = huffman.decompress_tensor(c, out=dec)
    t2 = time.perf_counter()
    print("prof=%s: encode %.4f ms/call, decode %.4f ms/call" % (prof, (t1 - t0) / 20 * 1e3, (t2 - t1) / 20 * 1e3))
    _lib.prof_get()
small = src[:4096].clone()
for prof in (False, True):
    _lib.prof_enable(prof); _lib.prof_reset()
    cs = huffman.compress_tensor(small, out=out)
    t0 = time.perf_counter()
    for _ in range(200): cs = huffman.compress_tensor(small, out=out)
    t1 = time.perf_counter()
    for _ in range(200): ds = huffman.decompress_tensor(cs, out=dec)
    t2 = time.perf_counter()
    print("4 KiB prof=%s: encode %.1f us/call, decode %.1f us/call" % (prof, (t1 - t0) / 200 * 1e6, (t2 - t1) / 200 * 1e6))
    _lib.prof_get()
