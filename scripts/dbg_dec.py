"""Debug helper: LZSS decode of small streams against the oracle, one case per line (flushes before each call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from raisin_amd import lz, _lib

def text(seed, n):
    import random
    rng = random.Random(seed)
    words = [b"the", b"quick", b"brown", b"fox", b"jumps", b"over", b"lazy", b"dog", b"compression", b"raisin"]
    out = bytearray()
    while len(out) < n:
        out += rng.choice(words) + b" "
    return bytes(out[:n])

cases = [("tiny", b"abcabcabcabcabcabcabcabc\n"), ("text 3000", text(1, 3000)), ("text 20000", text(2, 20000)), ("text 100000", text(3, 100000)),
         ("period 70000", bytes(np.random.default_rng(1).integers(97, 123, size=4096, dtype=np.uint8)) * 17),
         ("text 1M", text(4, 1 << 20)), ("zeros", b"\x00" * 300000), ("text 5M", text(5, 5 << 20))]
for name, data in cases:
    c = O.lzss_compress(data)
    print(name, len(data), "->", len(c), end=" ... ", flush=True)
    try:
        d = lz.Decompress(c)
        print("ok" if d == data else "MISMATCH at %d" % next((i for i in range(min(len(d), len(data))) if d[i] != data[i]), -1), len(d), flush=True)
    except Exception as e:
        print("EXC", e, flush=True)
