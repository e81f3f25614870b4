"""GPU parity: librsn Huffman encode (through the C ABI) vs the CPU oracle, bit-exact."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def huff():
    from raisin_amd import huffman
    return huffman


def rnd_bytes(seed, n, lo=0, hi=256):
    return np.random.default_rng(seed).integers(lo, hi, size=n, dtype=np.uint8).tobytes()


def fib_skewed(k=30):
    """k symbols with Fibonacci counts -> maximum code length k-1 (> 26: wide path)."""
    a, b, parts = 1, 1, []
    for i in range(k):
        parts.append(bytes([33 + i]) * a)
        a, b = b, a + b
    buf = bytearray(b"".join(parts))
    random.Random(5).shuffle(buf)
    return bytes(buf)


def test_fixtures_and_known_answers(huff, oracle, samiam, known):
    for data in (b"Hello world!\n", b"abc" * 8 + b"\n", samiam, (samiam * 20)[:65536], b"ab", b"a\nb\\", b"AB\\\\A"):
        assert huff.Compress(data) == oracle.huffman_compress(data)
    assert len(huff.Compress(b"Hello world!\n")) == known["reference"]["huffman_hello_size"]
    assert len(huff.Compress(b"abc" * 8 + b"\n")) == known["reference"]["huffman_abc_size"]
    assert huff.table(samiam) == oracle.huffman_table(samiam)


@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 255, 4095, 4096, 4097, 65535, 65536, 65537, 131072 + 5, 1 << 20, (1 << 22) + 12345])
def test_ascii_random_sizes(huff, oracle, n):
    data = rnd_bytes(n, n, 0, 128)
    assert huff.Compress(data) == oracle.huffman_compress(data)


@pytest.mark.parametrize("seed", range(6))
def test_ascii_skewed(huff, oracle, seed):
    rng = np.random.default_rng(100 + seed)
    k = int(rng.integers(2, 100))
    p = rng.dirichlet(np.ones(k) * 0.3)
    data = rng.choice(np.arange(20, 20 + k, dtype=np.uint8), size=300000 + seed * 7777, p=p).astype(np.uint8).tobytes()
    assert huff.Compress(data) == oracle.huffman_compress(data)


def test_single_symbol_and_empty(huff, oracle):
    from raisin_amd import RsnError
    assert huff.Compress(b"aaaa") == oracle.huffman_compress(b"aaaa") == b"4|a\\\n\x00"
    assert huff.Compress(b"z" * 100000) == oracle.huffman_compress(b"z" * 100000)
    with pytest.raises(RsnError):
        huff.Compress(b"")   # reference panics (huffman.go:102)


def test_wide_codes(huff, oracle):
    data = fib_skewed(30)
    t = oracle.huffman_table(data)
    assert max(x[3] for x in t) > 24
    assert huff.Compress(data) == oracle.huffman_compress(data)


@pytest.mark.parametrize("n", [1, 3, 17, 4096, 65537, 1 << 20])
def test_binary_rune_path(huff, oracle, n):
    data = rnd_bytes(7 * n + 1, n)   # invalid UTF-8 -> U+FFFD (huffman.go:309)
    assert huff.Compress(data) == oracle.huffman_compress(data)


def test_valid_utf8_text(huff, oracle):
    rng = random.Random(9)
    alphabet = "abc déf ✓ λ 𝄞 \n\\|0123"
    s = "".join(rng.choice(alphabet) for _ in range(200000)).encode("utf-8")
    assert huff.Compress(s) == oracle.huffman_compress(s)
    # truncated sequences at the very end and at tile boundaries
    for cut in (1, 2, 3):
        assert huff.Compress(s[:-cut]) == oracle.huffman_compress(s[:-cut])
    t = (s * 3)[:65536 * 2 + 1]
    assert huff.Compress(t) == oracle.huffman_compress(t)


def test_small_and_large_tiles(huff, oracle):
    """Inputs up to 2 MiB are cut into 4 KiB tiles, larger ones into 64 KiB tiles (huff_encode.hip SMALL_INPUT): the
    same bytes either side of the switch, on the byte, skewed-byte and rune paths, with multi-byte sequences across
    tile edges of both sizes and ragged last tiles."""
    rng = random.Random(77)
    alphabet = "ab déf ✓ λ 𝄞 \n\\|01"
    utf8 = "".join(rng.choice(alphabet) for _ in range(1 << 20)).encode("utf-8")
    for n in ((2 << 20) - 1, 2 << 20, (2 << 20) + 1, (2 << 20) + 4097, 4096 * 3 + 1, 4096 * 5 - 2):
        for data in (rnd_bytes(n, n, 0, 128), (utf8 * 3)[1:n + 1], rnd_bytes(n + 5, n)):
            c = huff.Compress(data)
            assert c == oracle.huffman_compress(data)
            assert huff.Decompress(c) == oracle.huffman_decompress(c)
    for n in (16 << 20, (16 << 20) + 1):                          # the rune path keeps the small tiles up to 16 MiB
        data = (utf8 * (n // len(utf8) + 2))[3:n + 3]
        assert huff.Compress(data) == oracle.huffman_compress(data)


def test_device_resident_api(huff, oracle):
    import torch
    data = rnd_bytes(3, 3 << 20, 0, 128)
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    out = huff.compress_tensor(src)
    assert bytes(out.cpu().numpy()) == oracle.huffman_compress(data)


def test_large_property_256MiB(huff, oracle):
    """Size-independent checks at a size the oracle still finishes in seconds."""
    import torch
    n = 1 << 28
    g = torch.Generator(device="cuda").manual_seed(0x5EED0002)
    src = torch.randint(0, 128, (n,), dtype=torch.uint8, device="cuda", generator=g)
    out = huff.compress_tensor(src)
    host = bytes(out.cpu().numpy())
    sep = host.index(b"\\\n")
    counts = torch.bincount(src.view(-1).to(torch.int64), minlength=128).cpu().numpy()
    ents, _ = oracle.header_entries(host)
    assert sorted(int(f) for f, _ in ents) == sorted(int(c) for c in counts if c)
    # all 128 symbols ~equiprobable -> every code is 7 bits
    assert len(host) - sep - 3 == n * 7 // 8 and host[sep + 2] == 0
    ref = oracle.huffman_compress(bytes(src.cpu().numpy()))
    assert host == ref


def test_rune_start_map_at_every_lane_and_stream_edge(huff, oracle):
    """The rune path classifies once (k_rune_hist) and the later passes take rune starts from the map: sequences of 2, 3 and 4 bytes
    straddling every 16-byte lane boundary, invalid bytes next to them, streams that end inside a sequence (Go: U+FFFD per byte,
    huffman.go:309) and right after one -- against the oracle."""
    pieces = ["\u00e9".encode(), "\u20ac".encode(), "\U0001F600".encode(), b"a", b"\x80", b"\xc2", b"\xe2\x82", b"\xf0\x9f\x98", b"\xff", b"\xed\xa0\x80", b"\xc0\x80"]
    cases = []
    for shift in range(0, 20):
        body = b"x" * shift + b"".join(pieces) * 9 + b"tail"
        for cut in (0, 1, 2, 3):
            cases.append(body + "\U0001F600".encode()[:4 - cut] if cut else body + "\U0001F600".encode())
    cases.append(("\u0416\u0443\u043a " * 5000).encode()[:-1])
    cases.append(bytes(range(256)) * 40)
    got = [huff.Compress(c) for c in cases]
    assert got == [oracle.huffman_compress(c) for c in cases]
