"""Ad-hoc timing of the device-resident Huffman codec on one workload (workloads.py name), per-kernel event timings."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import workloads as W
from raisin_amd import _lib, huffman


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "skewed"
    mib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    n = mib << 20
    src = W.config_input(kind, n, "cuda")
    c = huffman.compress_tensor(src)
    out = torch.empty(2 * n + (1 << 16), dtype=torch.uint8, device="cuda")
    d = huffman.decompress_tensor(c, out=out)
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    for name, fn in (("encode", lambda: huffman.compress_tensor(src)), ("decode", lambda: huffman.decompress_tensor(c, out=out))):
        _lib.prof_reset()
        t0 = time.perf_counter()
        r = fn()
        t1 = time.perf_counter()
        print("%s %s %d MiB: %.3f ms -> %d B" % (kind, name, mib, (t1 - t0) * 1e3, r.numel()))
        for k, (cnt, ms) in sorted(_lib.prof_get().items()):
            print("  %-22s %3d launches  %.3f ms total" % (k, cnt, ms))
    print("lossless:", bool(d.numel() == n and torch.equal(d, src)))


if __name__ == "__main__":
    main()
