"""Throughput against input size (device-resident calls): where a codec's time stops following the bytes, something in it is serial."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import workloads as W
from raisin_amd import huffman, lz

sizes = [int(a) << 20 for a in sys.argv[1:]] or [4 << 20, 16 << 20, 64 << 20, 256 << 20, 1024 << 20]
full = {k: W.config_input(k, max(sizes), "cuda") for k in ("2a", "skewed", "4", "3")}
for name, mod, key in (("huffman 2a", huffman, "2a"), ("huffman skewed", huffman, "skewed"), ("lzss text", lz, "4"), ("lzss periodic", lz, "3")):
    for n in sizes:
        d = full[key][:n].clone()
        c = mod.compress_tensor(d); o = mod.decompress_tensor(c); torch.cuda.synchronize()
        assert torch.equal(o, d)
        for _ in range(3): o = mod.decompress_tensor(c)   # (torch's allocator settles: the first gigabyte-sized outputs are fresh hipMallocs)
        torch.cuda.synchronize()
        reps = 5 if n >= (256 << 20) else 20
        t0 = time.perf_counter()
        for _ in range(reps): c = mod.compress_tensor(d)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(reps): o = mod.decompress_tensor(c)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        e, dd = (t1 - t0) / reps, (t2 - t1) / reps
        print("%-15s %5d MiB: encode %9.1f us (%8.1f GB/s)   decode %9.1f us (%8.1f GB/s)" % (name, n >> 20, e * 1e6, n / e / 1e9, dd * 1e6, n / dd / 1e9))
