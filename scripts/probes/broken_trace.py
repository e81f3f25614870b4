"""RSN_DEBUG trace of the LZSS encoder on a period broken every so often (the slowest shape): which stage decides"""
import sys, time, random
sys.path.insert(0, ".")
from raisin_amd import _lib, lz
rng = random.Random(5)
for p, n in ((256, 1 << 20), (37, 1 << 20), (256, 8 << 20)):
    unit = "".join(rng.choices("abcdefghijklmnopqrstuvwxyz ,.\n", k=p)).encode()
    b = bytearray((unit * (n // p + 1))[:n])
    at = 30000
    while at < n:
        b[at] = ord(rng.choice("ABCDEFG")); at += rng.randint(20000, 200000)
    data = bytes(b)
    lz.CompressAsync(data)
    sys.stderr.write("=== period %d, %d bytes\n" % (p, n)); sys.stderr.flush()
    _lib.prof_enable(True); _lib.prof_reset()
    t0 = time.perf_counter(); c = lz.CompressAsync(data); t = time.perf_counter() - t0
    pe = _lib.prof_get(); _lib.prof_enable(False)
    top = sorted(((v[1], k, v[0]) for k, v in pe.items()), reverse=True)[:8]
    sys.stderr.write("%.2f ms, %d -> %d B; %s\n" % (t * 1e3, n, len(c), ", ".join("%s x%d %.2f" % (k, m, ms) for ms, k, m in top))); sys.stderr.flush()
