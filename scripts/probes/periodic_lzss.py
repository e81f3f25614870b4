"""LZSS encode / decode time of periodic and mixed inputs (host-buffer calls, 1 MiB and 8 MiB): which shapes are out of line?
prints the cases slower than 4x the median of their size."""
import sys; sys.path.insert(0, ".")
import random
import numpy as np
from raisin_amd import _lib
import bench
L = _lib.lib()
rng = random.Random(21)
alph = "abcdefghijklmnopqrstuvwxyz ,.\n<>\\"
rows = []
for n in (1 << 20, 1 << 23):
    for period in (1, 2, 3, 5, 7, 16, 37, 100, 255, 256, 1000, 4095, 4096, 4097, 5000, 8192, 20000):
        for trial in range(2):
            k = rng.randint(1, min(len(alph), max(1, period)))
            letters = rng.sample(alph, k)
            unit = "".join(rng.choices(letters, k=period)).encode()
            data = (unit * (n // len(unit) + 1))[:n]
            if trial == 1:                                  # the period broken every so often
                b = bytearray(data)
                for at in range(rng.randint(1000, 50000), n, rng.randint(20000, 200000)):
                    b[at] = ord(rng.choice(alph))
                data = bytes(b)
            arr = np.frombuffer(data, dtype=np.uint8)
            try:
                c, _ = bench._host_call(L.rsn_lzss_compress, arr, 4096)
                te = sorted(bench._host_call(L.rsn_lzss_compress, arr, 4096)[1] for _ in range(3))[1]
                d, _ = bench._host_call(L.rsn_lzss_decompress, c)
                td = sorted(bench._host_call(L.rsn_lzss_decompress, c)[1] for _ in range(3))[1]
                ok = d.tobytes() == data
            except Exception as e:
                te = td = float("nan"); ok = False
            rows.append((n, period, trial, te * 1e3, td * 1e3, ok))
for n in (1 << 20, 1 << 23):
    sel = [r for r in rows if r[0] == n]
    me, md = float(np.median([r[3] for r in sel])), float(np.median([r[4] for r in sel]))
    print("n = %d: %d cases, encode median %.0f us max %.0f, decode median %.0f us max %.0f, mismatches %d" % (n, len(sel), me, max(r[3] for r in sel), md, max(r[4] for r in sel), sum(not r[5] for r in sel)))
    for r in sorted(sel, key=lambda r: -r[3])[:6]:
        if r[3] > 4 * me: print("   encode: period %5d trial %d: %9.0f us" % (r[1], r[2], r[3]))
    for r in sorted(sel, key=lambda r: -r[4])[:6]:
        if r[4] > 4 * md: print("   decode: period %5d trial %d: %9.0f us" % (r[1], r[2], r[4]))
