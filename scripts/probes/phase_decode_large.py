"""k_dec_phase at scale: device-resident Huffman round trips of periodic UTF-8 (the unit that does not settle at 4 MiB), 64 MiB ... 1 GiB"""
import sys, time; sys.path.insert(0, ".")
import torch
from raisin_amd import huffman, _lib
unit = b'a\xe4\xb8\x96\xc3\xa8\xe6\x9c\xac\xe4\xb8\x96u\xc3\xb6uuu\xe6\x9c\xac\xc3\xa8\xe6\x9c\xac'
u = torch.frombuffer(bytearray(unit), dtype=torch.uint8).cuda()
for mib, cut in ((4, 0), (4, 4), (64, 4), (64, 5), (256, 4), (256, 9), (1024, 4), (1024, 1)):      # (cut: bytes short of the size -- which inputs stick is a matter of alignment)
    n = (mib << 20) - cut
    src = u.repeat(n // len(unit) + 1)[:n].contiguous()
    c = huffman.compress_tensor(src)
    out = torch.empty(n + (1 << 20), dtype=torch.uint8, device="cuda")
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        d = huffman.decompress_tensor(c, out=out)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    _lib.prof_enable(True); _lib.prof_reset()
    d = huffman.decompress_tensor(c, out=out)
    p = _lib.prof_get(); _lib.prof_enable(False)
    print("%5d MiB - %d: decode %s ms, lossless %s, kernels %s" % (mib, cut, [round(x, 2) for x in ts], bool(torch.equal(d, src)), {k: (v[0], round(v[1], 2)) for k, v in p.items() if v[0] and v[1] > 0.05}), flush=True)
