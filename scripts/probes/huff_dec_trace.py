"""per-kernel times of one Huffman decode (device tensors) on a kind of scripts/probes/real_shapes.py: argv = MiB, kind"""
import sys, time
sys.path.insert(0, ".")
sys.argv = [sys.argv[0], sys.argv[1], sys.argv[2], "skip"]
import importlib.util, torch
spec = importlib.util.spec_from_file_location("rs", "scripts/probes/real_shapes.py")
src = open("scripts/probes/real_shapes.py").read().split("\ndef med(")[0]
g = {}
exec(compile(src, "real_shapes_head", "exec"), g)
from raisin_amd import _lib, huffman
for name, gen in g["kinds"].items():
    if sys.argv[2] not in name:
        continue
    data = gen(g["N"])
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    c = huffman.compress_tensor(t); huffman.decompress_tensor(c); torch.cuda.synchronize()
    _lib.prof_enable(True); _lib.prof_reset()
    t0 = time.perf_counter(); d = huffman.decompress_tensor(c); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pe = _lib.prof_get(); _lib.prof_enable(False)
    print("=== %s: decode %.2f ms (%d -> %d B)" % (name, dt * 1e3, c.numel(), d.numel()))
    for k, (cnt, ms) in sorted(pe.items(), key=lambda kv: -kv[1][1]):
        print("   %-24s x%-3d %.3f ms" % (k, cnt, ms))
