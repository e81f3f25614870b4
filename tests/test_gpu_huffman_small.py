"""GPU parity: the small-input Huffman path (huff_small.hip: host buffers up to 64 KiB, byte alphabets -- two launches to compress, one to
decompress) against the CPU oracle, bit-exact, and against the general path (the device-pointer entry points never take the small one)."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def huff():
    from raisin_amd import huffman
    return huffman


def _general_compress(huff, data):
    from raisin_amd import _lib
    if _lib.NO_TORCH:                                       # (tests/test_gpu_no_torch.py: this suite on the runtime a Go host gets)
        return _lib.dev_codec(_lib.lib().rsn_huffman_compress_dev, data, len(data) + len(data) // 8 + (1 << 16))
    import torch
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    return huff.compress_tensor(src).cpu().numpy().tobytes()


def _general_decompress(huff, stream):
    from raisin_amd import _lib
    if _lib.NO_TORCH:
        return _lib.dev_codec(_lib.lib().rsn_huffman_decompress_dev, stream, 8 * len(stream) + (1 << 16))
    import torch
    src = torch.frombuffer(bytearray(stream), dtype=torch.uint8).cuda()
    return huff.decompress_tensor(src).cpu().numpy().tobytes()


def _text(seed, n):
    rng = random.Random(seed)
    words = ["".join(rng.choice("etaoinshrdlucmfwypvbgkqjxz") for _ in range(rng.randint(1, 9))) for _ in range(300)]
    out = []
    size = 0
    while size < n:
        w = rng.choice(words) + rng.choice([" ", " ", " ", ", ", ".\n", "\\ "])
        out.append(w)
        size += len(w)
    return "".join(out).encode()[:n]


def _fib(k):
    a, b, parts = 1, 1, []
    for i in range(k):
        parts.append(bytes([40 + i]) * a)
        a, b = b, a + b
    buf = bytearray(b"".join(parts))
    random.Random(k).shuffle(buf)
    return bytes(buf)


def _inputs():
    rng = np.random.default_rng(0x5A11)
    yield "samiam-like text 64 KiB", _text(1, 65536)
    sam = open(os.path.join(os.path.dirname(__file__), "golden", "samiam.txt"), "rb").read()
    yield "the README's file repeated to 64 KiB (periodic: a wrong parse can last)", (sam * (65536 // len(sam) + 1))[:65536]
    yield "a period of 7 bytes", (b"abcabda" * 9400)[:65536]
    yield "uniform over 16 symbols: four phases", rng.integers(65, 81, size=60000, dtype=np.uint8).tobytes()
    yield "uniform over 32 symbols: five phases (the general decoder's)", rng.integers(65, 97, size=60000, dtype=np.uint8).tobytes()
    yield "the README's 13 bytes", b"Hello world!\n"
    yield "the README's 25 bytes", b"abcabcabcabcabcabcabcabc\n"
    for n in (2, 3, 5, 8, 15, 16, 17, 31, 33, 47, 63):                     # r06: the path from 2 bytes up (two distinct symbols at least)
        yield "tiny text %d" % n, (b"ab" + _text(n, n))[:n]
    for n in (64, 65, 100, 1000, 1023, 1024, 4096, 16384 + 3, 50000, 65535, 65536):
        yield "text %d" % n, _text(n, n)
        yield "uniform ascii %d" % n, rng.integers(0, 128, size=n, dtype=np.uint8).tobytes()
    yield "flat 64 symbols", rng.integers(32, 96, size=65536, dtype=np.uint8).tobytes()
    yield "two symbols", rng.integers(0, 2, size=40000, dtype=np.uint8).tobytes().replace(b"\x00", b"a").replace(b"\x01", b"\n")
    yield "two symbols, one rare", (b"a" * 65000 + b"\\" * 3 + b"a" * 533)
    yield "fibonacci counts, codes to 21 bits", _fib(22)
    yield "fibonacci counts, 16 symbols", _fib(16)
    yield "digits and separators", (b"12|3|\\\n45\\n|" * 4000)[:47001]
    k = 100
    p = rng.dirichlet(np.ones(k) * 0.2)
    yield "skewed 100 symbols", rng.choice(k, size=65536, p=p).astype(np.uint8).tobytes()
    yield "all 128 symbols", bytes(range(128)) * 512


@pytest.mark.parametrize("name,data", list(_inputs()), ids=[n for n, _ in _inputs()])
def test_small_path_is_the_oracle_and_the_general_path(huff, oracle, name, data):
    want = oracle.huffman_compress(data)
    got = huff.Compress(data)
    assert got == want
    if len(data) >= 16:                                                    # (device buffers want 16 bytes of input at least)
        assert _general_compress(huff, data) == want
    assert huff.Decompress(got) == data
    assert _general_decompress(huff, got) == data


def test_small_calls_are_two_launches_and_one(huff):
    from raisin_amd import _lib
    data = _text(7, 65536)
    huff.Compress(data)
    _lib.prof_enable(True)
    _lib.prof_reset()
    c = huff.Compress(data)
    enc = _lib.prof_get()
    _lib.prof_reset()
    assert huff.Decompress(c) == data
    dec = _lib.prof_get()
    _lib.prof_enable(False)
    assert {k: v[0] for k, v in enc.items() if v[0]} == {"huff_small_hist": 1, "huff_small_emit": 1}
    assert {k: v[0] for k, v in dec.items() if v[0]} == {"huff_small_dec": 1}
    # a flat code (128 symbols, seven bits each) has seven phases and none to find: its boundaries are arithmetic
    flat = np.random.default_rng(4).integers(0, 128, size=65536, dtype=np.uint8).tobytes()
    c = huff.Compress(flat)
    _lib.prof_enable(True)
    _lib.prof_reset()
    assert huff.Decompress(c) == flat
    dec = _lib.prof_get()
    _lib.prof_enable(False)
    assert {k: v[0] for k, v in dec.items() if v[0]} == {"huff_small_dec": 1}


def test_what_the_small_path_declines_takes_the_general_one(huff, oracle):
    from raisin_amd import _lib
    cases = [b"a" * 5000,                                  # one symbol
             "héllo wörld, ".encode() * 300,       # runes
             b"ab" * 20 + b"\xff" + b"ab" * 20,            # an invalid byte: U+FFFD
             b"x" * 63 + b"y",                             # at the lower size limit
             bytes(range(128)) * 513]                      # one chunk above 64 KiB
    for data in cases:
        c = huff.Compress(data)
        assert c == oracle.huffman_compress(data)
        assert huff.Decompress(c) == oracle.huffman_decompress(c)           # (one symbol: the reference's decoder returns it once, huffman.go:136-143)
    _lib.prof_enable(True)
    _lib.prof_reset()
    huff.Compress(cases[1])
    names = {k for k, v in _lib.prof_get().items() if v[0]}
    _lib.prof_enable(False)
    assert "huff_small_emit" not in names and "huff_emit_rune" in names


def _outcome(fn, *a):
    from raisin_amd import _lib
    try:
        return ("ok", fn(*a))
    except _lib.RsnError as e:
        return ("error", e.code)


def test_damaged_small_streams_behave_as_on_the_general_path(huff):
    rng = random.Random(11)
    data = _text(3, 30000)
    good = huff.Compress(data)
    sep = good.index(b"\\\n")
    streams = [good[:-1], good[:-7], good[:sep + 3], good[:sep + 2], good[:sep + 4],
               good[:sep + 2] + bytes([9]) + good[sep + 3:],                 # a pad of 9
               good[:sep + 2] + bytes([0]) + good[sep + 3:],
               good[:sep + 2] + bytes([7]) + good[sep + 3:],
               b"3|a2|b\\\n" + bytes([0]) + bytes(rng.randrange(256) for _ in range(500)),
               b"1|a\\\n\x00", b"5|a5|b", b"\\\n\x00\x00", b"9|a9|b\\\n\x03\xff\xff\xff",
               b"70000|a1|b\\\n\x00" + bytes(2000)]
    for k in range(12):                                                      # a flipped payload byte: decodes to something, or ends inside a codeword
        at = rng.randrange(sep + 3, len(good))
        streams.append(good[:at] + bytes([good[at] ^ (1 << rng.randrange(8))]) + good[at + 1:])
    for s in streams:
        assert _outcome(huff.Decompress, s) == _outcome(_general_decompress, huff, s), s[:40]


def test_foreign_headers_decode_as_the_reference_would(huff, oracle):
    """the counts only shape the tree (huffman.go:196-227): a header whose counts are not the payload's still decodes"""
    data = b"abracadabra, " * 700
    good = oracle.huffman_compress(data)
    sep = good.index(b"\\\n")
    # every count doubled: the same tree, a header that promises twice the symbols
    import re
    hdr = good[:sep]
    doubled = re.sub(rb"(\d+)\|", lambda m: str(2 * int(m.group(1))).encode() + b"|", hdr)
    s = doubled + good[sep:]
    want = oracle.huffman_decompress(s)
    assert huff.Decompress(s) == want == _general_decompress(huff, s)


def test_small_calls_from_many_threads(huff, oracle):
    """every thread has its own context: pinned staging, device scratch, the decoder's block flags and call numbers"""
    from concurrent.futures import ThreadPoolExecutor
    datas = [_text(100 + k, 3000 + 7919 * k % 60000) for k in range(8)]
    wants = [oracle.huffman_compress(d) for d in datas]

    def work(k):
        for _ in range(60):
            c = huff.Compress(datas[k])
            if c != wants[k] or huff.Decompress(c) != datas[k]:
                return False
        return True

    with ThreadPoolExecutor(max_workers=8) as ex:
        assert all(ex.map(work, range(8)))


def test_small_and_large_calls_interleave_on_one_thread(huff, oracle):
    """the small path polls flags instead of waiting for the stream: what follows on the same stream must still see its results in order"""
    big = _text(9, 3 << 20)
    small = _text(10, 20000)
    want_big, want_small = oracle.huffman_compress(big), oracle.huffman_compress(small)
    for _ in range(5):
        assert huff.Compress(small) == want_small
        assert huff.Compress(big) == want_big
        assert huff.Decompress(want_small) == small
        assert huff.Decompress(want_big) == big


def test_a_header_that_promises_less_than_the_payload_holds(huff, oracle):
    """the small-input decoder is offered a stream by its header's counts (at most 64 KiB of symbols); what the payload decodes to is
    another matter -- more than a block's stage, more than 64 KiB: the kernel says so and the general decoder runs"""
    import re
    data = _text(21, 100000)
    good = oracle.huffman_compress(data)
    assert len(good) < 65536 + 2048
    sep = good.index(b"\\\n")
    halved = re.sub(rb"(\d+)\|", lambda m: str(max(1, int(m.group(1)) // 2)).encode() + b"|", good[:sep])
    s = halved + good[sep:]
    want = oracle.huffman_decompress(s)
    assert len(want) > 65536
    assert huff.Decompress(s) == want == _general_decompress(huff, s)
