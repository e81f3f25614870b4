"""LZSS on a 96 MiB stream of alternating 1 MiB sections -- zeros, word text, 37-byte runs, noise, 200-byte records with edits, a short
period -- round trip, time and the chain walk's looks (RSN_LZSS_DEBUG=1)."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import workloads as W
from raisin_amd import lz, _lib
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(3)
sec = 1 << 20
txt = W.config_input("4", 16 * sec, dev)
parts = []
for i in range(16):
    rec = torch.randint(32, 127, (200,), device=dev, generator=g, dtype=torch.uint8).repeat(sec // 200 + 1)[:sec].clone()
    idx = torch.randint(0, sec, (sec // 50,), device=dev, generator=g)
    rec[idx] = torch.randint(32, 127, (sec // 50,), device=dev, generator=g, dtype=torch.uint8)
    parts += [torch.zeros(sec, dtype=torch.uint8, device=dev), txt[i * sec:(i + 1) * sec],
              torch.repeat_interleave(torch.randint(97, 101, (sec // 37 + 1,), device=dev, generator=g, dtype=torch.uint8), 37)[:sec],
              torch.randint(0, 256, (sec,), device=dev, generator=g, dtype=torch.uint8), rec,
              torch.randint(97, 123, (7,), device=dev, generator=g, dtype=torch.uint8).repeat(sec // 7 + 1)[:sec]]
d = torch.cat(parts).contiguous()
c = lz.compress_tensor(d); o = lz.decompress_tensor(c); torch.cuda.synchronize()
print("round trip", "ok" if torch.equal(o, d) else "MISMATCH", "ratio %.2f %%" % (100.0 * c.numel() / d.numel()))
_lib.prof_enable(True); _lib.prof_reset()
t0 = time.perf_counter(); c = lz.compress_tensor(d); torch.cuda.synchronize(); t1 = time.perf_counter()
p = _lib.prof_get(); _lib.prof_enable(False)
o = lz.decompress_tensor(c); torch.cuda.synchronize(); t2 = time.perf_counter()
print("mixed %d MiB: encode %.2f ms, decode %.2f ms" % (d.numel() >> 20, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
for k, (cnt, ms) in sorted(p.items()):
    if ms > 0.3: print("      %-24s %2d  %9.1f us" % (k, cnt, ms * 1e3))
