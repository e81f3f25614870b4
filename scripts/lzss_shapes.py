"""LZSS on inputs that are neither text nor noise nor 4096-periodic: runs, short periods, records with small edits -- round trip and time
(looking for cliffs: a shape that falls onto the sweep or the general parse for the whole stream)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from raisin_amd import lz, _lib

def shapes(n, dev):
    g = torch.Generator(device=dev); g.manual_seed(5)
    ar = torch.arange(n, device=dev)
    yield "zeros", torch.zeros(n, dtype=torch.uint8, device=dev)
    yield "period 2", (ar % 2 + 65).to(torch.uint8)
    yield "period 3", (ar % 3 + 65).to(torch.uint8)
    yield "period 1000", torch.randint(97, 123, (1000,), device=dev, generator=g, dtype=torch.uint8).repeat(n // 1000 + 1)[:n].contiguous()
    yield "period 5000", torch.randint(97, 123, (5000,), device=dev, generator=g, dtype=torch.uint8).repeat(n // 5000 + 1)[:n].contiguous()
    rec = torch.randint(32, 127, (200,), device=dev, generator=g, dtype=torch.uint8).repeat(n // 200 + 1)[:n].clone()
    idx = torch.randint(0, n, (n // 50,), device=dev, generator=g)
    rec[idx] = torch.randint(32, 127, (n // 50,), device=dev, generator=g, dtype=torch.uint8)
    yield "200-byte records, 2 % edits", rec
    yield "sawtooth 0..255", (ar % 256).to(torch.uint8)
    runs = torch.repeat_interleave(torch.randint(97, 101, (n // 37 + 1,), device=dev, generator=g, dtype=torch.uint8), 37)[:n].contiguous()
    yield "runs of 37", runs
    small = torch.randint(0, 300, (n // 4 + 1,), device=dev, generator=g, dtype=torch.int32)
    yield "int32 values below 300", small.view(torch.uint8)[:n].contiguous()
    yield "four-letter alphabet, random", (torch.randint(0, 4, (n,), device=dev, generator=g, dtype=torch.uint8) * 3 + 65)
    hexd = torch.randint(0, 16, (n,), device=dev, generator=g, dtype=torch.uint8)
    yield "hex digits, random", torch.where(hexd < 10, hexd + 48, hexd + 87)

for mib in [int(a) for a in sys.argv[1:]] or [16]:
    n = mib << 20
    for name, d in shapes(n, "cuda"):
        try:
            c = lz.compress_tensor(d); o = lz.decompress_tensor(c); torch.cuda.synchronize()
            ok = bool(torch.equal(o, d))
            t0 = time.perf_counter(); c = lz.compress_tensor(d); torch.cuda.synchronize(); t1 = time.perf_counter()
            o = lz.decompress_tensor(c); torch.cuda.synchronize(); t2 = time.perf_counter()
            print("%4d MiB %-28s encode %9.2f ms  decode %8.2f ms  ratio %6.2f %%  round trip %s" % (mib, name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, 100.0 * c.numel() / n, "ok" if ok else "MISMATCH"), flush=True)
        except Exception as e:      # noqa: BLE001
            print("%4d MiB %-28s %s: %s" % (mib, name, type(e).__name__, e), flush=True)
