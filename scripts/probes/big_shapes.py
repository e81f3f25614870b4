"""1 GiB of the run- and line-heavy kinds of real_shapes.py through the device API: encode / decode time, round trip (r06: the walk's
instance for repeated stretches at full size -- 131072 tiles, the serial in-tile parse, sections from 2 GiB up are the tests')."""
import sys, time
sys.path.insert(0, ".")
sys.argv = [sys.argv[0], "64", "", "skip"]
import torch
src_head = open("scripts/probes/real_shapes.py").read().split("\ndef med(")[0]
g = {}
exec(compile(src_head, "real_shapes_head", "exec"), g)
from raisin_amd import lz
for name in ("sparse: a byte in 100", "runs up to 1000", "csv, nine rows in ten the same", "a log line repeated 1-300 times", "word list, sorted", "text"):
    part = g["kinds"][name](64 << 20)
    t = torch.frombuffer(bytearray(part), dtype=torch.uint8).cuda().repeat(16)          # 1 GiB: the 64 MiB sixteen times (a period far beyond any window)
    z = lz.compress_tensor(t); torch.cuda.synchronize()
    t0 = time.perf_counter(); z = lz.compress_tensor(t); torch.cuda.synchronize(); te = time.perf_counter() - t0
    t0 = time.perf_counter(); u = lz.decompress_tensor(z); torch.cuda.synchronize(); td = time.perf_counter() - t0
    print("%-36s 1 GiB: encode %7.1f ms (%5.1f GB/s), decode %6.1f ms, ratio %5.1f %%, round trip %s" % (name, te * 1e3, t.numel() / te / 1e9, td * 1e3, 100.0 * z.numel() / t.numel(), "ok" if torch.equal(u, t) else "MISMATCH"), flush=True)
    del t, z, u
