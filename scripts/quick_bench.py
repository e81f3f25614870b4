"""Ad-hoc timing of the device-resident codecs with per-kernel event timings."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raisin_amd import _lib, huffman


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    hi = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    n = mib << 20
    g = torch.Generator(device="cuda").manual_seed(0x5EED0002)
    src = torch.randint(0, hi, (n,), dtype=torch.uint8, device="cuda", generator=g)
    out = torch.empty(huffman.compress_bound(n) if hi > 128 else n + (1 << 20), dtype=torch.uint8, device="cuda")
    for it in range(3):
        c = huffman.compress_tensor(src, out=out)
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    _lib.prof_reset()
    reps = 5
    t0 = time.perf_counter()
    for it in range(reps):
        c = huffman.compress_tensor(src, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("encode %d MiB (values < %d): %.3f ms  %.1f GB/s input, out %d B" % (mib, hi, dt * 1e3, n / dt / 1e9, c.numel()))
    for k, (cnt, ms) in sorted(_lib.prof_get().items()):
        print("  %-24s %3d launches  %.3f ms avg" % (k, cnt, ms / cnt))
    _lib.prof_enable(False)
    try:
        for it in range(2):
            d = huffman.decompress_tensor(c)
        torch.cuda.synchronize()
        _lib.prof_enable(True)
        _lib.prof_reset()
        t0 = time.perf_counter()
        for it in range(reps):
            d = huffman.decompress_tensor(c)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("decode: %.3f ms  %.1f GB/s output; equal=%s" % (dt * 1e3, d.numel() / dt / 1e9, bool(torch.equal(d, src)) if hi <= 128 else "n/a"))
        for k, (cnt, ms) in sorted(_lib.prof_get().items()):
            print("  %-24s %3d launches  %.3f ms avg" % (k, cnt, ms / cnt))
    except Exception as e:  # decode may not exist yet
        print("decode failed:", e)


if __name__ == "__main__":
    main()
