"""The Canterbury corpus' pi.txt from its definition: the first 1 000 000 decimal digits of pi, "31415926535..." without the
point.  The reference measured its LZSS engine on that file (/root/reference/ai/data.json, "name": "pi.txt", engine "lzss":
compressed_ratio 100.0, lossless false); tests/test_oracle.py::test_pi_txt_pins_the_legacy_encoder holds the oracle to that row.

The digits are generated, not committed (1 MB): mpmath's Chudnovsky series, about a minute on one core; the result is cached
under $TMPDIR and pinned by its sha256 (the same digest whichever way the digits are produced).

    python tests/golden/make_pi.py [out_file]
"""
import hashlib
import os
import sys
import tempfile

N_DIGITS = 1_000_000
SHA256 = "387877db67fdddbde761c053c4376e0b411b10fd2b126fd8b1249963cb628877"


def pi_digits(n=N_DIGITS):
    cache = os.path.join(tempfile.gettempdir(), "rsn_pi_%d.txt" % n)
    if os.path.exists(cache):
        d = open(cache, "rb").read()
        if len(d) == n and (n != N_DIGITS or hashlib.sha256(d).hexdigest() == SHA256):
            return d
    import mpmath
    mpmath.mp.dps = n + 20
    s = mpmath.nstr(mpmath.mp.pi, n + 10, strip_zeros=False)
    assert s[:2] == "3."
    d = ("3" + s[2:])[:n].encode()
    tmp = cache + ".%d" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(d)
    os.replace(tmp, cache)
    return d


if __name__ == "__main__":
    d = pi_digits()
    print(len(d), hashlib.sha256(d).hexdigest(), d[:32].decode())
    if len(sys.argv) > 1:
        open(sys.argv[1], "wb").write(d)
