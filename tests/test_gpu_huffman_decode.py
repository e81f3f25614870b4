"""GPU parity: librsn Huffman decode (through the C ABI) vs the CPU oracle, bit-exact."""
import ctypes
import random

import numpy as np
import pytest

from test_gpu_huffman_encode import fib_skewed, rnd_bytes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def huff():
    from raisin_amd import huffman
    return huffman


def test_fixtures(huff, oracle, samiam):
    for data in (b"Hello world!\n", b"abc" * 8 + b"\n", samiam, (samiam * 20)[:65536], b"ab", b"a\nb\\", b"AB\\\\A", b"1|2||33|\n\n7"):
        c = oracle.huffman_compress(data)
        assert huff.Decompress(c) == oracle.huffman_decompress(c) == data
    # reference round trip (cli_test.go:33-40) entirely on the device
    assert huff.Decompress(huff.Compress(samiam)) == samiam


def test_single_symbol_quirk(huff, oracle):
    c = oracle.huffman_compress(b"aaaa")
    assert huff.Decompress(c) == oracle.huffman_decompress(c) == b"a"   # huffman.go:136-143
    c = oracle.huffman_compress("ééé".encode())
    assert huff.Decompress(c) == "é".encode()


@pytest.mark.parametrize("n", [2, 3, 15, 16, 17, 255, 4095, 4097, 8191, 8192, 8193, 65537, 1 << 20, (1 << 22) + 12345])
def test_flat_alphabet_sizes(huff, oracle, n):
    data = rnd_bytes(n, n, 0, 128)
    c = oracle.huffman_compress(data)
    assert huff.Decompress(c) == data


@pytest.mark.parametrize("seed", range(8))
def test_skewed_self_sync(huff, oracle, seed):
    rng = np.random.default_rng(200 + seed)
    k = int(rng.integers(2, 120))
    p = rng.dirichlet(np.ones(k) * (0.05 + 0.2 * seed))
    data = rng.choice(np.arange(8, 8 + k, dtype=np.uint8), size=400000 + seed * 9991, p=p).astype(np.uint8).tobytes()
    c = oracle.huffman_compress(data)
    assert huff.Decompress(c) == data


@pytest.mark.parametrize("L", [1, 2, 3, 4, 5, 6, 7])
def test_flat_codes_every_width(huff, oracle, L):
    """Exactly 2^L equiprobable symbols -> all codes L bits -> the fixed-width unpack path."""
    k = 1 << L
    for reps, extra in ((37, 0), (8200, 0), (1000, 3)):
        lo = 0 if k == 128 else 40
        buf = bytearray(bytes(range(lo, lo + k)) * reps)
        random.Random(L * 100 + reps).shuffle(buf)
        data = bytes(buf)
        if extra:   # perturb counts a little: lengths stay equal while no count doubles another
            data += bytes(range(lo, lo + min(k, extra)))
        c = oracle.huffman_compress(data)
        t = oracle.huffman_table(data)
        assert {x[3] for x in t} == {L}
        assert huff.Decompress(c) == data
        assert huff.Compress(data) == c            # flat-code emit kernel, every width and ragged tails
        assert huff.Decompress(huff.Compress(data)) == data


def test_two_symbols_one_bit_codes(huff, oracle):
    data = rnd_bytes(1, 700001, 0, 2)
    c = oracle.huffman_compress(data)
    assert huff.Decompress(c) == data


def test_wide_codes_long_walk(huff, oracle):
    data = fib_skewed(30)
    c = oracle.huffman_compress(data)
    assert huff.Decompress(c) == data


@pytest.mark.parametrize("n", [1, 3, 17, 4096, 65537, 1 << 19])
def test_binary_is_lossy_like_the_reference(huff, oracle, n):
    data = rnd_bytes(7 * n + 1, n)
    c = oracle.huffman_compress(data)
    want = oracle.huffman_decompress(c)
    assert huff.Decompress(c) == want
    assert huff.Decompress(huff.Compress(data)) == want


def test_utf8_text(huff, oracle):
    rng = random.Random(9)
    s = "".join(rng.choice("abc déf ✓ λ 𝄞 \n\\|0123") for _ in range(150000)).encode("utf-8")
    c = oracle.huffman_compress(s)
    assert huff.Decompress(c) == s


def test_header_order_is_free(huff, oracle, samiam):
    """Any entry order is a legal reference output (Go map order, huffman.go:312)."""
    c = oracle.huffman_compress(samiam)
    ents, rest = oracle.header_entries(c)
    rng = random.Random(4)
    for _ in range(5):
        rng.shuffle(ents)
        while ents[-1][1] == b"\\":
            rng.shuffle(ents)
        shuffled = b"".join(f + b"|" + s for f, s in ents) + b"\\\n" + rest
        assert huff.Decompress(shuffled) == samiam == oracle.huffman_decompress(shuffled)


def test_large_pad_byte(huff, oracle, samiam):
    c = oracle.huffman_compress(samiam)
    sep = c.index(b"\\\n")
    pad = c[sep + 2]
    weird = c[:sep + 2] + bytes([pad + 16]) + b"\x00\x00" + c[sep + 3:]
    assert oracle.huffman_decompress(weird) == samiam
    assert huff.Decompress(weird) == samiam


def test_errors_where_the_reference_panics(huff, oracle, samiam):
    from raisin_amd import RsnError
    good = oracle.huffman_compress(samiam)
    bad_inputs = [
        b"no separator at all",
        b"2|\\\\\n\x00",                      # '\\' is the last header entry (huffman.go:210)
        b"3|a2|b\\\n\x09\x00",               # pad exceeds payload bits (huffman.go:294)
        b"4|a\\\n\x00\xff",                  # single-leaf tree with payload (huffman.go:139-140)
        b"3|a2|b1|c\\\n\x00",                # multi-symbol tree, empty payload (huffman.go:145)
        b"\\\n\x00\x00",                     # empty header (huffman.go:102)
        good[:-1] + bytes([good[-1] ^ 1]) if False else good[:-2],   # truncated: ends inside a codeword (usually)
    ]
    for b in bad_inputs:
        try:
            want = oracle.huffman_decompress(b)
        except oracle.OracleError:
            want = None
        if want is None:
            with pytest.raises(RsnError):
                huff.Decompress(b)
        else:
            assert huff.Decompress(b) == want


def test_size_query_checks_the_stream_and_cannot_overflow(huff):
    """the size query (d_out NULL) is answered from the header -- after the cheap format checks, so that a malformed stream
    is refused by the query itself, and with counts that exceed what the payload can hold (a foreign header; a count past 2^63 - 1
    reads as MaxInt64 like strconv.Atoi, huffman.go:207) answered by the payload's own bound instead of a wrapped sum."""
    import torch
    from raisin_amd import _lib

    def query(stream):
        t = torch.frombuffer(bytearray(stream + b"\0" * 16), dtype=torch.uint8).cuda()
        got = ctypes.c_size_t(0)
        rc = _lib.lib().rsn_huffman_decompress_dev(t.data_ptr(), len(stream), None, 0, ctypes.byref(got), None)
        return rc, got.value

    good = huff.Compress(b"abracadabra" * 1000)
    rc, need = query(good)
    assert rc == -7 and 11000 <= need <= 11000 + 32
    assert query(b"4|a\\\n\x00\x80")[0] == -3                   # single-symbol tree with a payload (reference recurses without end)
    assert query(b"1|a1|b\\\n\x09\x80")[0] == -3                # pad exceeds the payload
    assert query(b"1|a1|b")[0] == -3                               # no separator
    huge = b"99999999999999999999999|a" + b"18446744073709551615|b" + b"\\\n\x00" + bytes(range(100))
    rc, need = query(huge)
    assert rc == -7 and need <= 4 * 800 + 32                       # 100 payload bytes: at most 800 symbols of at most 4 bytes
    assert huff.Decompress(huge) == bytes(b"ab"[(b >> (7 - k)) & 1] for b in range(100) for k in range(8))


def test_a_buffer_that_is_too_small_is_never_overrun(huff):
    """the emit pass is queued before the host has seen the total (one sync per call), so a too-small buffer may be partly
    written before RSN_ERR_CAPACITY comes back (rsn.h says so) -- but never past out_cap: a canary behind the buffer stays, for the
    general decoder (skewed codes, multi-block), the flat one, and a stream whose payload ends inside a codeword (RSN_ERR_FORMAT)."""
    import torch
    from raisin_amd import _lib
    rng = np.random.default_rng(7)
    skew = bytes((rng.geometric(0.25, size=1 << 20).clip(max=60) + 32).astype(np.uint8))
    flat = bytes(rng.integers(0, 128, size=1 << 20, dtype=np.uint8))
    for data in (skew, flat):
        comp = huff.Compress(data)
        src = torch.frombuffer(bytearray(comp + b"\0" * 64), dtype=torch.uint8).cuda()
        for cap in (16, 4096, 65536 + 16, len(data) // 2 // 16 * 16, len(data) - 16 - len(data) % 16):
            buf = torch.full((cap + 4096,), 0xA5, dtype=torch.uint8, device="cuda")
            got = ctypes.c_size_t(0)
            rc = _lib.lib().rsn_huffman_decompress_dev(src.data_ptr(), len(comp), buf.data_ptr(), cap, ctypes.byref(got), None)
            torch.cuda.synchronize()
            assert rc == _lib.RSN_ERR_CAPACITY and got.value >= len(data), (rc, cap)
            assert bool((buf[cap:] == 0xA5).all()), cap
        # a payload cut inside a codeword: FORMAT or a shorter output, never a write past the buffer
        cut = comp[:len(comp) - len(comp) // 3]
        src2 = torch.frombuffer(bytearray(cut + b"\0" * 64), dtype=torch.uint8).cuda()
        cap = len(data) // 2 // 16 * 16
        buf = torch.full((cap + 4096,), 0xA5, dtype=torch.uint8, device="cuda")
        got = ctypes.c_size_t(0)
        rc = _lib.lib().rsn_huffman_decompress_dev(src2.data_ptr(), len(cut), buf.data_ptr(), cap, ctypes.byref(got), None)
        torch.cuda.synchronize()
        assert rc in (_lib.RSN_ERR_CAPACITY, -3), rc
        assert bool((buf[cap:] == 0xA5).all())


def test_device_resident_round_trip_256MiB(huff):
    import torch
    n = 1 << 28
    g = torch.Generator(device="cuda").manual_seed(0x5EED0002)
    src = torch.randint(0, 128, (n,), dtype=torch.uint8, device="cuda", generator=g)
    c = huff.compress_tensor(src)
    d = huff.decompress_tensor(c)
    assert d.numel() == n and torch.equal(d, src)
    # skewed (variable-length codes, real self-synchronisation) at size
    w = torch.tensor([2.0 ** (-i / 3) for i in range(90)], device="cuda")
    src2 = (torch.multinomial(w, 1 << 26, replacement=True).to(torch.uint8) + 32).contiguous()
    c2 = huff.compress_tensor(src2)
    d2 = huff.decompress_tensor(c2)
    assert torch.equal(d2, src2)


def test_second_level_tables_and_switches(oracle):
    """Codes longer than the first-level table: sub-tables in LDS (byte alphabets), through L2 (rune alphabets with tens of
    thousands of symbols), and the bit-by-bit walk below them (Fibonacci counts: 29-bit codes).  The same streams decode
    the same with RSN_NO_MULTI=1 (one codeword per lookup: the second formulation of the walks) and RSN_DEC_WARM=0 (no warm-up:
    most blocks' guessed entries are wrong and the fixing passes do the work): separate processes, the switches are read once."""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import hashlib\n"
            "from raisin_amd import huffman\n"
            "from oracle import oracle\n"
            "from test_gpu_huffman_encode import fib_skewed, rnd_bytes\n"
            "import workloads as W\n"
            "for d in (fib_skewed(30), bytes(W.skewed_bytes(3 << 20).numpy()), rnd_bytes(11, 3 << 20), bytes(W.config_input('4', 3 << 20).numpy()), b'ab' * 70000 + b'c'):\n"
            "    c = oracle.huffman_compress(d)\n"
            "    print(hashlib.sha256(huffman.Decompress(c)).hexdigest(), hashlib.sha256(oracle.huffman_decompress(c)).hexdigest())\n"
            ) % (root, os.path.join(root, "tests"))
    outs = []
    for env in ({}, {"RSN_NO_MULTI": "1"}, {"RSN_DEC_WARM": "0"}, {"RSN_DEC_WARM": "0", "RSN_NO_MULTI": "1"}):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l.split() for l in r.stdout.strip().splitlines()]
        assert len(lines) == 5 and all(a == b for a, b in lines), (env, lines)
        outs.append(lines)
    assert all(o == outs[0] for o in outs)


def test_a_code_that_does_not_self_synchronise(huff, oracle):
    """Code lengths {3 x 7, 6 x 8}: every codeword's length is a multiple of three, so a decode that starts on the wrong residue never
    finds the boundaries again (an all-even code would not do: the pad keeps the payload's parity).  Two lanes in three start their
    warm-up on a wrong residue, the in-block fixed point takes hundreds of rounds, every block's guessed entry is as likely wrong as
    not and so is its predecessor's first exit: r04 went through pass after pass -- the loop behind the speculative D2 + D3 -- until
    the parse had crossed the stream (55 passes); r05 calls such a block stuck and decodes the stream from every entry of every lane
    (k_dec_phase, see test_streams_that_do_not_settle_are_decoded_from_every_entry); the bytes are the oracle's."""
    rng = np.random.default_rng(77)
    syms = np.arange(65, 80, dtype=np.uint8)
    w = np.array([8] * 7 + [1] * 8, dtype=np.float64)
    for n in (300000, 300001, 1 << 20):
        body = syms[rng.choice(15, size=n, p=w / w.sum())].tobytes()
        t = oracle.huffman_table(body)
        assert sorted(x[3] for x in t) == [3] * 7 + [6] * 8, sorted(x[3] for x in t)
        c = oracle.huffman_compress(body)
        assert huff.Compress(body) == c
        assert huff.Decompress(c) == body == oracle.huffman_decompress(c)


def test_streams_that_do_not_settle_are_decoded_from_every_entry(huff, oracle):
    """Periodic data parses in more than one phase for as long as the period lasts, and a code whose lengths share a factor never finds
    the boundaries again from a wrong residue: the fixed point's rounds then walk the stretch a lane at a time, its passes a block at a
    time (r05: 474 ms for the first of these inputs).  A block that is still handing corrections on after SYNC_ROUNDS rounds is called
    stuck and the stream is decoded from every possible entry of every lane (k_dec_phase): the oracle's bytes, in milliseconds."""
    import time
    from raisin_amd import _lib
    unit = b'a\xe4\xb8\x96\xc3\xa8\xe6\x9c\xac\xe4\xb8\x96u\xc3\xb6uuu\xe6\x9c\xac\xc3\xa8\xe6\x9c\xac'
    rng = np.random.default_rng(2)
    five = np.repeat(np.arange(63) + 48, [32] * 31 + [1] * 32)                   # code lengths 5 and 10: five residues
    cases = {"a UTF-8 unit of 13 runes repeated to 4 MiB": (unit * ((4 << 20) // len(unit) + 1))[:4 << 20],
             "the same, 64 KiB (the small-input decoder declines runes)": ("héllo wörld 世界 " * 5000).encode()[:65536],
             "lengths 5 and 10, 2 MiB": rng.choice(five, size=2 << 20).astype(np.uint8).tobytes(),
             "lengths 3 and 6, 1 MiB + 1": np.arange(65, 80, dtype=np.uint8)[rng.choice(15, size=(1 << 20) + 1, p=np.array([8] * 7 + [1] * 8) / 64.0)].tobytes()}
    settled = 0
    for name, data in cases.items():
        c = oracle.huffman_compress(data)
        assert huff.Compress(data) == c, name
        want = oracle.huffman_decompress(c)
        huff.Decompress(c)                                                       # (warm: arenas, the result block)
        _lib.prof_enable(True)
        _lib.prof_reset()
        t0 = time.perf_counter()
        got = huff.Decompress(c)
        dt = time.perf_counter() - t0
        p = _lib.prof_get()
        _lib.prof_enable(False)
        assert got == want, name
        assert dt < 0.1, (name, dt)
        settled += p.get("huff_dec_phase", (0, 0))[0] > 0
    assert settled >= 2, "no input reached k_dec_phase"
