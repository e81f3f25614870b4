import sys; sys.path.insert(0, ".")
import random
import numpy as np
from raisin_amd import _lib
import bench
L = _lib.lib()
rng = random.Random(7)
alph = "abcdefghijklmnopqrstuvwxyz ,.\n"
n, period = 1 << 20, 37
unit = "".join(rng.choices(alph, k=period)).encode()
b = bytearray((unit * (n // len(unit) + 1))[:n])
at = rng.randint(1000, 50000)
brk = []
while at < n:
    b[at] = ord(rng.choice(alph)); brk.append(at); at += rng.randint(20000, 200000)
print("breaks at tiles", [x // 8192 for x in brk], "offsets", [x % 8192 for x in brk])
arr = np.frombuffer(bytes(b), dtype=np.uint8)
c, t = bench._host_call(L.rsn_lzss_compress, arr, 4096)
