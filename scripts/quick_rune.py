"""Huffman on realistic multi-byte UTF-8 text (rune path, huffman.go:309): skewed Cyrillic letters, spaces, some ASCII."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from raisin_amd import _lib, huffman


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rng = np.random.default_rng(7)
    letters = [chr(c) for c in range(0x430, 0x450)] + [" ", " ", " ", ",", ".", "\n", "a", "e", "€", "𝄞"]
    p = np.array([2.0 ** (-i / 5) for i in range(len(letters))]); p /= p.sum()
    n_runes = (mib << 20) // 2
    idx = rng.choice(len(letters), size=n_runes, p=p)
    enc = [l.encode("utf-8") for l in letters]
    data = b"".join(enc[i] for i in idx[: 1 << 20])
    reps = (mib << 20) // len(data) + 1
    buf = (data * reps)[: mib << 20]
    while buf and (buf[-1] & 0xC0) == 0x80:      # do not cut a rune
        buf = buf[:-1]
    if buf and buf[-1] >= 0xC0:
        buf = buf[:-1]
    src = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
    n = src.numel()
    out = torch.empty(n + n // 4 + (1 << 20), dtype=torch.uint8, device="cuda")
    dec = torch.empty(n + (1 << 20), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        c = huffman.compress_tensor(src, out=out)
        d = huffman.decompress_tensor(c, out=dec)
    _lib.prof_enable(True); _lib.prof_reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c = huffman.compress_tensor(src, out=out)
    t1 = time.perf_counter()
    d = huffman.decompress_tensor(c, out=dec)
    t2 = time.perf_counter()
    print("utf8 text %d MiB: enc %.2f ms, dec %.2f ms, ratio %.2f%%, lossless=%s" % (mib, (t1 - t0) * 1e3, (t2 - t1) * 1e3, 100.0 * c.numel() / n, bool(torch.equal(d, src))))
    for k, (cnt, ms) in sorted(_lib.prof_get().items()):
        print("    %-24s x%-3d %.3f ms total" % (k, cnt, ms))


if __name__ == "__main__":
    main()
