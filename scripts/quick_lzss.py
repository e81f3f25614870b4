"""Ad-hoc timing of the device-resident LZSS codec with per-kernel event timings."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from raisin_amd import _lib, lz


def gen(kind, n):
    if kind == "period":      # config 3: one 4096-byte block without 5C/FF, repeated
        rng = np.random.default_rng(0x5EED0003)
        vals = np.array([v for v in range(256) if v not in (0x5C, 0xFF)], dtype=np.uint8)
        blk = vals[rng.integers(0, len(vals), size=4096)]
        return torch.from_numpy(np.tile(blk, n // 4096)).cuda()
    if kind == "text":        # config 4: Zipf words
        rng = np.random.default_rng(0x5EED0004)
        vocab = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(4096)]
        ranks = rng.zipf(1.3, size=n // 4) % 4096
        buf = b" ".join(vocab[r] for r in ranks)[:n]
        return torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
    if kind == "text1":       # the same vocabulary with the classic exponent 1 (p ~ 1/rank): a flatter word distribution
        rng = np.random.default_rng(0x5EED0004)
        vocab = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(4096)]
        p = 1.0 / np.arange(1, 4097)
        ranks = rng.choice(4096, size=n // 4, p=p / p.sum())
        buf = b" ".join(vocab[r] for r in ranks)[:n]
        return torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    return torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "text"
    mib = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    src = gen(kind, mib << 20)
    n = src.numel()
    c = lz.compress_tensor(src)
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    _lib.prof_reset()
    t0 = time.perf_counter()
    c = lz.compress_tensor(src)
    t1 = time.perf_counter()
    print("%s %d MiB: encode %.1f ms (%.2f GB/s) -> %d B (%.2f%%)" % (kind, mib, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9, c.numel(), 100.0 * c.numel() / n))
    for k, (cnt, ms) in sorted(_lib.prof_get().items()):
        print("  %-20s %3d launches  %.3f ms total" % (k, cnt, ms))
    out = torch.empty(n + (1 << 16), dtype=torch.uint8, device="cuda")
    d = lz.decompress_tensor(c, out=out)
    torch.cuda.synchronize()
    _lib.prof_reset()
    t0 = time.perf_counter()
    d = lz.decompress_tensor(c, out=out)
    t1 = time.perf_counter()
    print("decode %.1f ms (%.2f GB/s) lossless=%s" % ((t1 - t0) * 1e3, n / (t1 - t0) / 1e9, bool(torch.equal(d, src))))
    for k, (cnt, ms) in sorted(_lib.prof_get().items()):
        print("  %-20s %3d launches  %.3f ms total" % (k, cnt, ms))


if __name__ == "__main__":
    main()
