"""Config 2b: huffman on uniform bytes 0x00-0xFF (rune path: Go UTF-8 semantics, lossy like the reference)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raisin_amd import _lib, huffman

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = mib << 20
g = torch.Generator(device="cuda").manual_seed(0x5EED0002)
src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g)
out = torch.empty(huffman.compress_bound(n), dtype=torch.uint8, device="cuda")
c = huffman.compress_tensor(src, out=out)
_lib.prof_enable(True)
_lib.prof_reset()
t0 = time.perf_counter()
c = huffman.compress_tensor(src, out=out)
t1 = time.perf_counter()
d = huffman.decompress_tensor(c)
t2 = time.perf_counter()
print("2b %d MiB: enc %.2f ms, dec %.2f ms, out %d B (%.2f%%), decoded %d B" % (mib, (t1 - t0) * 1e3, (t2 - t1) * 1e3, c.numel(), 100.0 * c.numel() / n, d.numel()))
for k, (cnt, ms) in sorted(_lib.prof_get().items()):
    print("    %-24s x%-3d %.3f ms total" % (k, cnt, ms))
if mib <= 64:
    from oracle import oracle as O
    ref = O.huffman_compress(bytes(src.cpu().numpy()))
    print("bit-exact vs oracle:", bytes(c.cpu().numpy()) == ref, " decode ==", bytes(d.cpu().numpy()) == O.huffman_decompress(ref))
