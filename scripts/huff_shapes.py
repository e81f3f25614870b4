"""Huffman on skewed and degenerate alphabets: round trip and time (looking for cliffs)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from raisin_amd import huffman

def shapes(n, dev):
    g = torch.Generator(device=dev); g.manual_seed(9)
    u = torch.rand(n, device=dev, generator=g)
    yield "one symbol", torch.full((n,), 65, dtype=torch.uint8, device=dev)
    yield "two symbols, equal", (torch.randint(0, 2, (n,), device=dev, generator=g, dtype=torch.uint8) + 65)
    yield "99.9 % one symbol, 95 others", torch.where(u < 0.999, torch.full((n,), 32, dtype=torch.uint8, device=dev), (u * 1e6).to(torch.int64).remainder(95).to(torch.uint8) + 33)
    yield "geometric, 64 symbols (p halves)", torch.clamp((-torch.log2(u.clamp_min(1e-12))).to(torch.uint8), max=63) + 40
    yield "uniform 3 symbols", (torch.randint(0, 3, (n,), device=dev, generator=g, dtype=torch.uint8) + 65)
    yield "latin-1 text-like bytes (invalid UTF-8 -> U+FFFD)", torch.where(u < 0.9, (u * 1e5).to(torch.int64).remainder(26).to(torch.uint8) + 97, (u * 1e7).to(torch.int64).remainder(64).to(torch.uint8) + 192)

for mib in [int(a) for a in sys.argv[1:]] or [256]:
    n = mib << 20
    for name, d in shapes(n, "cuda"):
        try:
            c = huffman.compress_tensor(d); o = huffman.decompress_tensor(c); torch.cuda.synchronize()
            same = o.numel() == d.numel() and bool(torch.equal(o, d))
            t0 = time.perf_counter(); c = huffman.compress_tensor(d); torch.cuda.synchronize(); t1 = time.perf_counter()
            o = huffman.decompress_tensor(c); torch.cuda.synchronize(); t2 = time.perf_counter()
            print("%4d MiB %-50s encode %8.2f ms  decode %8.2f ms  ratio %6.2f %%  %s" % (mib, name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, 100.0 * c.numel() / n,
                  "lossless" if same else "decoded %d bytes (lossy like the reference)" % o.numel()), flush=True)
        except Exception as e:      # noqa: BLE001
            print("%4d MiB %-50s %s: %s" % (mib, name, type(e).__name__, e), flush=True)
