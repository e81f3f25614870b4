"""a longer fuzz than the suite's for round 5's new Huffman paths: small inputs (huff_small.hip) and periodic / repetitive inputs of a few
hundred KiB to a few MiB (k_dec_phase when they do not settle), all against the oracle; damaged small streams against the general path."""
import sys, time; sys.path.insert(0, ".")
import random
import numpy as np
import torch
from raisin_amd import huffman, _lib
from oracle import oracle as O
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = random.Random(seed)
nprng = np.random.default_rng(seed)

def gen_small():
    n = rng.choice([rng.randint(64, 300), rng.randint(300, 5000), rng.randint(5000, 65536), 65536, 65535, 4096])
    kind = rng.randrange(6)
    if kind == 0:
        k = rng.randint(2, 128); p = nprng.dirichlet(np.ones(k) * rng.choice([0.05, 0.3, 1.0, 10.0]))
        return nprng.choice(k, size=n, p=p).astype(np.uint8).tobytes()
    if kind == 1:
        unit = bytes(rng.randrange(128) for _ in range(rng.randint(1, 200)))
        return (unit * (n // len(unit) + 1))[:n]
    if kind == 2:
        unit = bytes(rng.choice(b"ab") for _ in range(rng.randint(1, 40)))
        b = bytearray((unit * (n // len(unit) + 1))[:n])
        for _ in range(rng.randint(0, 5)): b[rng.randrange(n)] = rng.randrange(128)
        return bytes(b)
    if kind == 3:
        a, c, parts = 1, 1, []
        for i in range(rng.randint(3, 22)):
            parts.append(bytes([33 + i]) * a); a, c = c, a + c
        buf = bytearray(b"".join(parts)); rng.shuffle(buf); return bytes(buf[:65536]) if len(buf) >= 64 else bytes(buf) * 8
    if kind == 4:
        L = rng.randint(1, 7); return nprng.integers(0, 1 << L, size=n, dtype=np.uint8).tobytes()
    return bytes(rng.choice(b"the quick brown fox\n\\|0123456789") for _ in range(n))

def gen_periodic():
    n = rng.choice([1 << 18, 1 << 20, (1 << 21) + rng.randint(0, 99), 1 << 22])
    alph = rng.choice(["abcdefghijklmnopqrstuvwxyz ,.\n", "aeiouäöüßéèñ世界日本語 ", "01", "abcdefgh"])
    p = rng.choice([2, 3, 5, 7, 11, 13, 17, 20, 29, 40, 64, 100, 257])
    k = rng.randint(2, min(len(alph), max(2, p)))
    letters = rng.sample(alph, k)
    unit = "".join(rng.choices(letters, [rng.random() ** 3 + 0.01 for _ in letters], k=p)).encode()
    return (unit * (n // len(unit) + 1))[:n]

def general_decompress(stream):
    src = torch.frombuffer(bytearray(stream), dtype=torch.uint8).cuda()
    return huffman.decompress_tensor(src).cpu().numpy().tobytes()

t0 = time.time(); n_small = n_per = n_dmg = 0
while time.time() - t0 < budget:
    data = gen_small()
    if len(data) < 1: continue
    want = O.huffman_compress(data)
    got = huffman.Compress(data)
    assert got == want, ("small compress", seed, n_small, len(data))
    assert huffman.Decompress(got) == O.huffman_decompress(want), ("small decompress", seed, n_small, len(data))
    n_small += 1
    if n_small % 4 == 0 and len(want) > 40:
        s = bytearray(want); at = rng.randrange(len(s)); s[at] ^= 1 << rng.randrange(8); s = bytes(s)
        def outcome(fn, x):
            try: return ("ok", fn(x))
            except _lib.RsnError as e: return ("error", e.code)
        assert outcome(huffman.Decompress, s) == outcome(general_decompress, s), ("damaged", seed, n_dmg, at)
        n_dmg += 1
    if n_small % 25 == 0:
        data = gen_periodic()
        c = huffman.Compress(data)
        t1 = time.time()
        d = huffman.Decompress(c)
        dt = time.time() - t1
        assert d == O.huffman_decompress(c), ("periodic", seed, n_per, len(data))
        assert dt < 0.5, ("periodic decode slow", seed, n_per, len(data), dt)
        n_per += 1
print("seed %d: %d small inputs, %d damaged streams, %d periodic inputs in %.0f s: all as the oracle" % (seed, n_small, n_dmg, n_per, time.time() - t0))
