"""Timing of huffman encode/decode on flat (2a) and skewed inputs, with per-kernel event timings."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raisin_amd import _lib, huffman


def run(name, src, reps=5):
    n = src.numel()
    out = torch.empty(n + n // 4 + (1 << 20), dtype=torch.uint8, device="cuda")
    dec = torch.empty(n + (1 << 20), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        c = huffman.compress_tensor(src, out=out)
        d = huffman.decompress_tensor(c, out=dec)
    _lib.prof_enable(True)
    _lib.prof_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        c = huffman.compress_tensor(src, out=out)
    t1 = time.perf_counter()
    for _ in range(reps):
        d = huffman.decompress_tensor(c, out=dec)
    t2 = time.perf_counter()
    prof = _lib.prof_get()
    _lib.prof_enable(False)
    print("%s: %d MiB  enc %.3f ms  dec %.3f ms  ratio %.2f%%  lossless=%s" % (
        name, n >> 20, (t1 - t0) / reps * 1e3, (t2 - t1) / reps * 1e3, 100.0 * c.numel() / n, bool(torch.equal(d, src))))
    for k, (cnt, ms) in sorted(prof.items()):
        print("    %-22s x%-3d %.3f ms avg" % (k, cnt // reps if cnt >= reps else cnt, ms / cnt))


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n = mib << 20
    g = torch.Generator(device="cuda").manual_seed(0x5EED0002)
    flat = torch.randint(0, 128, (n,), dtype=torch.uint8, device="cuda", generator=g)
    run("flat-2a", flat)
    if os.environ.get("RSN_NO_FLAT") is None and len(sys.argv) > 2:
        return
    w = torch.tensor([2.0 ** (-i / 6) for i in range(96)], device="cuda")
    sk = (torch.multinomial(w, n // 4, replacement=True).to(torch.uint8) + 32).repeat(4).contiguous()
    run("skewed-text-like", sk)


if __name__ == "__main__":
    main()
