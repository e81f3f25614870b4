"""Huffman decode time of periodic inputs (host-buffer call, 1 MiB and 4 MiB): which periods / alphabets send the synchronisation through
many passes?  prints the cases slower than 3x the median of their size."""
import sys; sys.path.insert(0, ".")
import random
import numpy as np
from raisin_amd import _lib
import bench
L = _lib.lib()
rng = random.Random(11)
rows = []
alph_ascii = "abcdefghijklmnopqrstuvwxyz ,.\n"
alph_utf8 = "aeiouäöüßéèñ世界日本語 "
for n in (1 << 20, 1 << 22):
    for kind, alph in (("ascii", alph_ascii), ("utf8", alph_utf8)):
        for period in (3, 5, 7, 11, 13, 17, 20, 29, 40, 64, 100, 257, 1000):
            for trial in range(2):
                k = rng.randint(2, min(len(alph), period))
                letters = rng.sample(alph, k)
                weights = [rng.random() ** 3 + 0.01 for _ in letters]
                unit = "".join(rng.choices(letters, weights, k=period)).encode()
                data = (unit * (n // len(unit) + 1))[:n]
                arr = np.frombuffer(data, dtype=np.uint8)
                try:
                    c, _ = bench._host_call(L.rsn_huffman_compress, arr)
                    ts = []
                    for _ in range(3):
                        d, t = bench._host_call(L.rsn_huffman_decompress, c)
                        ts.append(t)
                    rows.append((n, kind, period, trial, sorted(ts)[1] * 1e3, d.size))
                except Exception as e:
                    rows.append((n, kind, period, trial, float("nan"), -1))
for n in (1 << 20, 1 << 22):
    sel = [r for r in rows if r[0] == n and r[4] == r[4]]
    med = float(np.median([r[4] for r in sel]))
    slow = [r for r in sel if r[4] > 3 * med]
    print("n = %d: %d cases, median %.0f us, max %.0f us, %d slower than 3x the median" % (n, len(sel), med, max(r[4] for r in sel), len(slow)))
    for r in sorted(slow, key=lambda r: -r[4])[:12]:
        print("   %-5s period %4d trial %d: %9.0f us" % (r[1], r[2], r[3], r[4]))
