"""LZSS encode of short periodic / nearly periodic / mixed inputs against the oracle (round 5 changed when stretches are placed by
arithmetic and how the sweep cuts its strips); windows 4096 and others."""
import sys, time; sys.path.insert(0, ".")
import random
from raisin_amd import lz
from oracle import oracle as O
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = random.Random(seed)
alph = "abcdefghijklmnopqrstuvwxyz ,.\n<>\\\xff"
def gen():
    n = rng.choice([rng.randint(100, 9000), rng.randint(9000, 70000), rng.randint(70000, 200000)])
    p = rng.choice([1, 2, 3, 7, 37, 100, 255, 256, 1000, 3461, 4095, 4096, 4097, 5000, 9000])
    k = rng.randint(1, len(alph))
    unit = "".join(rng.choices(rng.sample(alph, k), k=p)).encode("latin-1")
    b = bytearray((unit * (n // len(unit) + 1))[:n])
    for _ in range(rng.choice([0, 0, 1, 3, 10])):
        b[rng.randrange(n)] = ord(rng.choice(alph))
    if rng.random() < 0.3:
        cut = rng.randrange(n); b[cut:cut] = bytes(rng.randrange(256) for _ in range(rng.randint(1, 3000)))
    return bytes(b)
t0 = time.time(); cases = 0
while time.time() - t0 < budget:
    data = gen()
    w = rng.choice([4096, 4096, 4096, 300, 1024, 8192])
    c = lz.CompressAsync(data, False, w)
    assert c == O.lzss_compress(data, w), ("lzss", seed, cases, len(data), w)
    assert lz.Decompress(c) == data
    cases += 1
print("seed %d: %d inputs in %.0f s: all as the oracle" % (seed, cases, time.time() - t0))
