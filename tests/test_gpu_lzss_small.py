"""GPU parity: the small-input LZSS path (lzss_small.hip: host buffers up to 2 KiB -- the reference's own table is files of 13-25 bytes,
README.md:153-167 -- one launch of one block each way) against the CPU oracle, bit-exact, and against the general path (the
device-pointer entry points never take the small one)."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lz():
    from raisin_amd import lz
    return lz


def _general(fn_name, data, cap, *extra):
    from raisin_amd import _lib
    return _lib.dev_codec(getattr(_lib.lib(), fn_name), data, cap, *extra)


def _cases():
    rng = random.Random(2048)
    out = [b"Hello world!\n", b"abcabcabcabcabcabcabcabc\n", b"a", b"ab", b"aaaa", b"<", b"\\", b"\xff", b"<<<<\\\\\xff\xff<", b"abcdeabcde", b"abcdefabcdef",
           b"abcdefg1abcdefg2abcdefg3", bytes(range(256)), b"\x00" * 2048, b"ab" * 1024, b"xyz" * 600, bytes([92]) * 1024, bytes([255]) * 1000, b"<" * 2048]
    words = [b"the", b"quick", b"brown", b"fox", b"<tag>", b"back\\slash", b"\xff\xfe", b"a", b"I", b"compression"]
    for n in (13, 25, 64, 100, 257, 1000, 2047, 2048):
        t = bytearray()
        while len(t) < n:
            t += rng.choice(words) + rng.choice([b" ", b"\n", b", "])
        out.append(bytes(t[:n]))
    for alphabet, n in ((b"ab", 2048), (b"abc<\\\xff", 1500), (bytes(range(256)), 2048), (b"\\<", 1024), (b"ACGT", 2000)):
        out.append(bytes(rng.choice(alphabet) for _ in range(n)))
    return out


def test_small_lzss_compress_is_the_oracles(lz, oracle):
    for data in _cases():
        for w in (4096, 16, 100, 0, 1):
            want = oracle.lzss_compress(data, w)
            assert lz.CompressAsync(data, False, w) == want, (data[:40], len(data), w)
        c = lz.CompressAsync(data)
        if len(data) >= 16:
            assert _general("rsn_lzss_compress_dev", data, 2 * len(data) + 4096, 4096) == c, data[:40]
        assert lz.Decompress(c) == data, (data[:40], len(data))


def test_small_lzss_decompress(lz, oracle):
    """Streams the reference's encoder writes, and hand-written ones: chains of tokens that copy tokens, zero-length tokens, a token of
    8192 bytes, escapes whose 5C runs cross token seams, a dangling escape; what the path does not take (malformed tokens, pointers
    before the data, outputs above 8 KiB) comes back from the general path with its error or its bytes."""
    from raisin_amd import RsnError
    streams = [oracle.lzss_compress(d, w) for d in _cases() for w in (4096, 7)]
    streams += [b"a<1,1><1,1><2,2><4,4><8,8>", b"abc<3,3><6,6><12,12>x<1,1>", b"ab<0,0>c<2,0>", b"x" + b"<1,1>" * 400, b"0123456789" * 5 + b"<50,50>" * 40,
                b"\\\\\\a<1,1>", b"\\", b"a\\", b"\\\xff\xff\\\\<3,2>", b"q" * 2000 + b"<2000,2000><4000,4000>", b"lit>,1<2,1>>"]
    for c in streams:
        want = oracle.lzss_decompress(c)
        assert lz.Decompress(c) == want, c[:60]
        if len(c) >= 16:
            assert _general("rsn_lzss_decompress_dev", c, len(want) + 64) == want, c[:60]
    big = b"q" * 2000 + b"<2000,2000>" * 4                      # 10 000 bytes: above what the block expands -- the general path's bytes
    assert lz.Decompress(big) == oracle.lzss_decompress(big)
    for bad, code in ((b"ab<9,2>", -3), (b"<1,1>", -3), (b"abc<2,3>", -3), (b"abc<x,1>", -3), (b"ab<1<1,1>", -3)):
        with pytest.raises(RsnError) as e:
            lz.Decompress(bad)
        assert e.value.code == code, bad


def test_small_lzss_is_one_launch_each_way(lz):
    from raisin_amd import _lib
    data = b"the quick brown fox jumps over the lazy dog, the quick brown fox\n" * 8
    _lib.prof_enable(True)
    _lib.prof_reset()
    c = lz.CompressAsync(data)
    pe = _lib.prof_get()
    _lib.prof_reset()
    assert lz.Decompress(c) == data
    pd = _lib.prof_get()
    _lib.prof_enable(False)
    assert {k: v[0] for k, v in pe.items()} == {"lzss_small_enc": 1} and {k: v[0] for k, v in pd.items()} == {"lzss_small_dec": 1}, (pe, pd)


def test_small_lzss_from_eight_threads(lz, oracle):
    import threading
    cases = _cases()
    want = [oracle.lzss_compress(d) for d in cases]
    errors = []

    def work(t):
        for r in range(20):
            for i, d in enumerate(cases):
                if (i + t + r) % 3:
                    continue
                c = lz.CompressAsync(d)
                if c != want[i] or lz.Decompress(c) != d:
                    errors.append((t, i))
    ts = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:5]
