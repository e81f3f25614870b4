"""Config 4: -algorithm=lzss,huffman layered on synthetic text, every layer timed (device-resident)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raisin_amd import huffman, lz
from quick_lzss import gen


def timed(fn, *a, **k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn(*a, **k)
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    src = gen("text", mib << 20)
    n = src.numel()
    b1 = torch.empty(2 * n + (1 << 16), dtype=torch.uint8, device="cuda")
    b2 = torch.empty(2 * n + (1 << 16), dtype=torch.uint8, device="cuda")
    b3 = torch.empty(2 * n + (1 << 16), dtype=torch.uint8, device="cuda")
    b4 = torch.empty(n + (1 << 16), dtype=torch.uint8, device="cuda")
    for rep in range(2):                                   # engine.go:443-452 (compress), :454-479 (decompress, reverse order)
        l1, t_lz = timed(lz.compress_tensor, src, out=b1)
        l2, t_hf = timed(huffman.compress_tensor, l1, out=b2)
        d1, t_hd = timed(huffman.decompress_tensor, l2, out=b3)
        d0, t_ld = timed(lz.decompress_tensor, d1, out=b4)
    ok = bool(torch.equal(d0, src))
    tot = t_lz + t_hf + t_hd + t_ld
    print("lzss,huffman on %d MiB text: lzss enc %.1f ms, huffman enc %.2f ms, huffman dec %.2f ms, lzss dec %.1f ms; total %.1f ms = %.0f MB/s; "
          "ratio %.2f%% (lzss alone %.2f%%) lossless=%s" % (mib, t_lz, t_hf, t_hd, t_ld, tot, n / tot / 1e3, 100.0 * l2.numel() / n, 100.0 * l1.numel() / n, ok))


if __name__ == "__main__":
    main()
