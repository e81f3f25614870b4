"""GPU parity against the SECOND restatement (oracle/literal.py: huffman.go / lzss.go the way the Go reads, independent of the C oracle):
the library's bytes on small inputs -- the sizes of the reference's own table -- must be the literal restatement's, both codecs, both
directions, the layered form included."""
import random

import pytest

pytestmark = pytest.mark.gpu


def test_library_against_the_literal_restatement():
    from oracle import literal as L
    from raisin_amd import huffman, lz
    rng = random.Random(606)
    alphabets = [b"ab", b"abcdefgh \n", bytes(range(32, 127)), bytes(range(256)), b"<\\\xff>,0123", "äöü€𝄞 ab".encode(), b"\xe2\x82\xac\xe2\x82xy\xc3"]
    for it in range(120):
        alph = alphabets[it % len(alphabets)]
        n = rng.choice((2, 3, 13, 25, 64, 100, 257, 500))
        data = bytes(rng.choice(alph) for _ in range(n))
        for w in (4096, 16, 0):
            c = lz.CompressAsync(data, False, w)
            assert c == L.lzss_compress(data, w), (it, w, data[:30])
        assert lz.Decompress(c) == L.lzss_decompress(c) == data
        if len({r for _, r in L.go_runes(data)}) > 1:
            h = huffman.Compress(data)
            assert h == L.huffman_compress(data), (it, data[:30])
            assert huffman.Decompress(h) == L.huffman_decompress(h)
            layered = huffman.Compress(lz.CompressAsync(data))                    # `-algorithm=lzss,huffman` (engine.go:443-452)
            if len(set(lz.CompressAsync(data))) > 1:
                assert layered == L.huffman_compress(L.lzss_compress(data))
    for name in ("Hello world!\n", "abcabcabcabcabcabcabcabc\n"):                 # README.md:153-167
        d = name.encode()
        assert lz.CompressAsync(d) == L.lzss_compress(d) and huffman.Compress(d) == L.huffman_compress(d)
