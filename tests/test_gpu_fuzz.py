"""GPU differential fuzzing: random streams through librsn and the CPU oracle must agree byte for
byte -- or both must reject.  Seeds are fixed; sizes keep the oracle within seconds."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MORE = int(os.environ.get("RSN_FUZZ", "1"))      # RSN_FUZZ=20 runs twenty times the seeds


@pytest.fixture(scope="module")
def mods():
    from raisin_amd import RsnError, huffman, lz
    return lz, huffman, RsnError


def _token_stream(rng, n_items, *, max_ptr, bad=0.0):
    """A hand-built LZSS stream: literal runs (never '<') and "<ptr,len>" tokens that are valid for
    the decoder (len <= ptr <= bytes produced so far), with odd but legal spellings mixed in; with
    probability `bad` an item is an out-of-range token (both sides reject) or a spelling the
    reference's error-dropping Atoi lets through but librsn refuses by design.
    Returns (stream, has_loose_spelling)."""
    lit = bytes(b for b in range(256) if b != 0x3C)
    out = bytearray()
    produced = 0
    loose = False
    for _ in range(n_items):
        r = rng.random()
        if produced == 0 or r < 0.45:
            k = rng.choice((1, 1, 2, 3, 7, 15, 16, 17, 40, 300))
            run = bytes(rng.choice(b">,0123456789\\\xff" if rng.random() < 0.3 else lit) for _ in range(k))
            out += run
            produced += k
            continue
        if rng.random() < bad:
            if rng.random() < 0.5:
                out += rng.choice([b"<%d,1>" % (produced + 1), b"<2,3>", b"<4294967296,1>", b"<99999999999,1>", b"<5,99999999999>"])
            else:
                out += rng.choice([b"<", b"<12", b"<12,", b"<,3>", b"<3,>", b"<1x,1>", b"<12,3<", b"<+3,1>", b"<3,0x>", b"<00000000003,1>"])
                loose = True
            continue
        ptr = rng.randint(1, min(produced, max_ptr))
        ln = rng.choice((0, 1, 2, ptr, rng.randint(0, ptr), min(ptr, rng.randint(0, 40))))
        if produced > 300000:                                  # "<p,p>" doubles the output: keep the stream bounded
            ln = min(ln, 40)
        spelling = rng.random()
        if spelling < 0.1:
            tok = b"<%010d,%010d>" % (ptr, ln)                 # the longest legal token: 23 bytes
        elif spelling < 0.2:
            tok = b"<0%d,00%d>" % (ptr, ln)
        else:
            tok = b"<%d,%d>" % (ptr, ln)
        out += tok
        produced += ln
    return bytes(out), loose


@pytest.mark.parametrize("seed", range(12 * MORE))
def test_lzss_decode_fuzz(mods, oracle, seed):
    lz, _, RsnError = mods
    rng = random.Random(1000 + seed)
    for case in range(6):
        max_ptr = rng.choice((3, 40, 4096, 9000, 16384, 40000, 1 << 30))
        bad = 0.0 if case < 4 else 0.02
        data, loose = _token_stream(rng, rng.choice((5, 60, 800, 6000)), max_ptr=max_ptr, bad=bad)
        try:
            want = oracle.lzss_decompress(data)
        except oracle.OracleError:
            want = None
        try:
            got = lz.Decompress(data)
        except RsnError:
            got = None
        if got is None:
            assert want is None or loose, (seed, case, max_ptr, len(data))      # refusing is only allowed where documented
        else:
            assert got == want, (seed, case, max_ptr, len(data))


def _structured(rng, n):
    """Bytes with repeats at many scales: the encoder's bucket search, its hand-back to the sweep
    and the greedy chain all get exercised."""
    alpha = rng.choice([b"ab", b"abc", b"abcdefgh <\\\xff", bytes(range(32, 127)), bytes(range(256))])
    pieces = [bytes(rng.choice(alpha) for _ in range(rng.choice((1, 2, 3, 5, 8, 13, 40, 300, 2000)))) for _ in range(12)]
    out = bytearray()
    while len(out) < n:
        r = rng.random()
        if r < 0.5:
            out += rng.choice(pieces)
        elif r < 0.8:
            out += bytes(rng.choice(alpha) for _ in range(rng.randint(1, 30)))
        elif out:
            back = rng.randint(1, min(len(out), 5000))
            k = rng.randint(1, min(back, 600))
            out += out[len(out) - back:len(out) - back + k]
        else:
            out += b"x"
    return bytes(out[:n])


@pytest.mark.parametrize("seed", range(10 * MORE))
def test_lzss_encode_fuzz(mods, oracle, seed):
    lz, _, _ = mods
    rng = random.Random(2000 + seed)
    for _ in range(3):
        data = _structured(rng, rng.choice((200, 5000, 17000, 33000)))
        w = rng.choice((4096, 4096, 4096, 1, 2, 17, 100, 1000, 4095, 5000, 8192))
        c = lz.CompressAsync(data, False, w)
        assert c == oracle.lzss_compress(data, w), (seed, len(data), w)
        assert lz.Decompress(c) == data


@pytest.mark.parametrize("seed", range(8 * MORE))
def test_huffman_fuzz(mods, oracle, seed):
    _, huffman, _ = mods
    rng = random.Random(3000 + seed)
    np_rng = np.random.default_rng(3000 + seed)
    for _ in range(4):
        k = rng.choice((1, 2, 3, 5, 17, 60, 128, 200, 256))
        n = rng.choice((1, 2, 9, 257, 5000, 70000, 300001))
        syms = rng.sample(range(256), k)
        p = np_rng.dirichlet(np.full(k, rng.choice((0.05, 0.3, 1.0, 10.0))))
        data = np.array(syms, dtype=np.uint8)[np_rng.choice(k, size=n, p=p)].tobytes()
        c = huffman.Compress(data)
        ref = oracle.huffman_compress(data)
        assert c == ref, (seed, k, n)
        try:
            want = oracle.huffman_decompress(c)
        except oracle.OracleError:
            want = None
        if want is not None:
            assert huffman.Decompress(c) == want


@pytest.mark.parametrize("seed", range(6 * MORE))
def test_huffman_rune_fuzz(mods, oracle, seed):
    """Rune path (huffman.go:309): valid 2/3/4-byte sequences, thousands of distinct sparse runes,
    invalid and truncated sequences (each U+FFFD for ONE byte), all mixed."""
    _, huffman, _ = mods
    rng = random.Random(4000 + seed)
    pools = {
        "ascii": [chr(c) for c in range(32, 127)] + ["\n"],
        "two": [chr(c) for c in range(0x80, 0x800, rng.choice((1, 7)))],
        "cjk": [chr(rng.randrange(0x4E00, 0xA000)) for _ in range(rng.choice((20, 3000)))],
        "four": [chr(rng.randrange(0x10000, 0x110000)) for _ in range(rng.choice((3, 400)))],
        "hot": ["\u201c", "\u201d", "\u2014", "\u20ac"],
    }
    bad = [b"\x80", b"\xbf", b"\xc0\x80", b"\xc1", b"\xf5", b"\xff", b"\xe0\x80\x80", b"\xed\xa0\x80", b"\xf0\x80\x80\x80",
           b"\xf4\x90\x80\x80", b"\xe1\x80", b"\xf1\x80\x80", b"\xc2"]
    weights = {"ascii": rng.choice((0, 5)), "two": 4, "cjk": rng.choice((0, 3)), "four": 1, "hot": 2}
    names = [k for k, w in weights.items() for _ in range(w)]
    out = bytearray()
    n = rng.choice((50, 3000, 120000, 400000))
    while len(out) < n:
        if rng.random() < 0.02:
            out += rng.choice(bad)
        else:
            out += rng.choice(pools[rng.choice(names)]).encode("utf-8")
    data = bytes(out)
    c = huffman.Compress(data)
    assert c == oracle.huffman_compress(data), (seed, n)
    try:
        want = oracle.huffman_decompress(c)
    except oracle.OracleError:
        want = None
    if want is not None:
        assert huffman.Decompress(c) == want


@pytest.mark.parametrize("seed", range(6 * MORE))
def test_fuzz_sharded_huffman_stream(mods, oracle, seed):
    """rsn_huffman_compress_sharded on random inputs and slice counts: bytes == the single call's == the oracle's.  Inputs mix ASCII,
    valid multi-byte sequences and invalid bytes, so that cuts fall next to (and would fall inside) sequences of every length and
    slices differ in whether they hold runes at all."""
    lz, huffman, RsnError = mods
    rng = random.Random(9000 + seed)
    pieces = []
    target = rng.choice((70, 300, 5000, 70000, 400000))
    size = 0
    while size < target:
        kind = rng.random()
        if kind < 0.35:
            p = bytes(rng.choice(b"abcdefgh \n\\|0123456789") for _ in range(rng.randint(1, 400)))
        elif kind < 0.6:
            p = "".join(rng.choice("é€𝄞ßж中🙂") for _ in range(rng.randint(1, 120))).encode()
        elif kind < 0.8:
            p = bytes(rng.randrange(256) for _ in range(rng.randint(1, 200)))          # mostly invalid sequences
        else:
            p = bytes([rng.choice((0xE2, 0xF0, 0xC3, 0x82, 0xAC))]) * rng.randint(1, 9)  # lead bytes and continuation bytes on their own
        pieces.append(p)
        size += len(p)
    data = b"".join(pieces)
    ref = oracle.huffman_compress(data)
    assert huffman.Compress(data) == ref
    for G in (2, rng.randint(3, 9), rng.randint(10, 40)):
        assert huffman.CompressSharded(data, G) == ref, (seed, G, len(data))


@pytest.mark.parametrize("seed", range(10 * MORE))
def test_periodic_tail_fuzz(mods, oracle, seed):
    """r06's arithmetic paths under fuzz: a random head (text, noise, nothing), then a block of random length repeated under a window
    that is or is not its length (only W-periodic data takes the paths; the rest must not), random remainders, sometimes a changed byte
    somewhere (the tail then begins behind it, or is too short to be taken), sometimes bytes that need an escape (the encoder's path
    declines).  Encode == the oracle's bytes; decode == the input; and the oracle's stream + a token run a foreign encoder might append."""
    lz, _, _ = mods
    rng = random.Random(9000 + seed)
    W = rng.choice((4096, 4096, 4096, 2048, 1024, 256, 4080, 16))
    period = rng.choice((W, W, W, W // 2 if W >= 32 else W, W + 16, 3 * W // 4 if W >= 64 else W))
    alphabet = bytes(b for b in range(256) if b not in (0x5C, 0xFF)) if rng.random() < 0.8 else bytes(range(256))
    blk = bytes(rng.choice(alphabet) for _ in range(period))
    head = rng.choice((b"", bytes(rng.choice(alphabet) for _ in range(rng.randint(1, 30000))),
                       b" ".join(rng.choice((b"the", b"quick", b"<b>", b"fox", b"lazy")) for _ in range(rng.randint(10, 20000)))))
    reps = rng.randint(40, 1600000 // max(period, 1) + 40)                          # (up to 1.6 MB: the decoder takes a token run from 1 MiB of output up)
    body = bytearray(blk * reps + blk[: rng.choice((0, 0, 1, 3, 9, 10, 11, period // 2, period - 1))])
    if rng.random() < 0.3:
        body[rng.randrange(len(body))] ^= 0x20
    data = head + bytes(body)
    want = oracle.lzss_compress(data, W)
    got = lz.CompressAsync(data, False, W)
    assert got == want, (seed, W, period, len(head), len(data))
    assert lz.Decompress(got) == data
    # a foreign tail: the oracle's stream + one token repeated (P = any distance inside the data) + sometimes a last item
    P = rng.choice((1, 5, 255, 256, 4096, 8192, min(len(data), 6000)))
    if P <= len(data):
        extra = b"<%d,%d>" % (P, P) * rng.randint(2, 1300000 // (P + 1) + 2) + rng.choice((b"", b"<%d,%d>" % (P, rng.randint(0, P)), b"end\xff."))
        stream = want + extra
        assert lz.Decompress(stream) == oracle.lzss_decompress(stream), (seed, P, len(extra))


def _stretches(rng, n):
    """Units of 1 to 200 bytes repeated 1 to 400 times, between them now and then a few bytes of anything, a stretch of text, a copy of an
    earlier stretch's unit shifted by a byte or cut short -- what chain_period_visit's keys are worked out from (r06)."""
    alphabets = [b"ab", b"0,.\n", b"abcdefghijklmnopqrstuvwxyz \n", bytes(range(32, 127)), b"\x00\x01\x02\xfe", b"<>\\a"]
    out = bytearray()
    units = []
    while len(out) < n:
        r = rng.random()
        if r < 0.6 or not units:
            al = rng.choice(alphabets)
            p = rng.choice((1, 1, 2, 3, 4, 5, 7, 8, 11, 12, 16, 17, 31, 33, 63, 64, 65, 100, 191, 192, 193, 200))
            u = bytes(rng.choice(al) for _ in range(p))
            units.append(u)
        elif r < 0.8:
            u = rng.choice(units)                                    # the same unit again: a stretch further back that ends differently
        else:
            u = rng.choice(units)
            k = rng.randrange(len(u))
            u = u[k:] + u[:k] if rng.random() < 0.5 else u[: max(1, len(u) - 1)]   # a rotation of it, or the unit a byte shorter
        reps = rng.choice((1, 2, 3, 5, 9, 30, 100, 400))
        body = u * reps
        cut = rng.randrange(len(u)) if rng.random() < 0.5 else 0  # the stretch ends inside a unit
        out += body[: len(body) - cut]
        r2 = rng.random()
        if r2 < 0.3:
            out += bytes(rng.choice(rng.choice(alphabets)) for _ in range(rng.randint(1, 9)))
        elif r2 < 0.4:
            out += bytes(rng.choice(alphabets[2]) for _ in range(rng.randint(50, 3000)))
    return bytes(out[:n])


@pytest.mark.parametrize("seed", range(16 * MORE))
def test_stretches_of_short_periods_fuzz(mods, oracle, seed):
    """r06: the walk's arithmetic for positions inside a stretch of a short period against the oracle's search, on streams made of such
    stretches -- windows 4096, 1024 and 300, so that stretches begin before the window, inside it and at its edge."""
    lz, _, _ = mods
    rng = random.Random(9000 + seed)
    data = _stretches(rng, rng.choice((40000, 150000, 400000)))
    for w in ((4096,) if seed % 4 else (4096, 1024, 300)):
        c = lz.CompressAsync(data, False, w)
        assert c == oracle.lzss_compress(data, w), "seed %d, window %d" % (seed, w)
    assert lz.Decompress(c) == data
