"""CPU tests pinning the oracle (oracle/*.c) against every known answer the
reference publishes (README sizes), the survey's independent hashes and the
reference's own round-trip tests (lzss_test.go:25-47, cli_test.go:33-40)."""
import hashlib
import random

import pytest


def sha(b):
    return hashlib.sha256(b).hexdigest()


HELLO = b"Hello world!\n"
ABC = b"abc" * 8 + b"\n"


def test_fixture_is_the_references(samiam, known):
    assert len(samiam) == 3461
    assert sha(samiam) == known["survey"]["samiam_sha256"]


def test_readme_known_answer_sizes(oracle, known):
    ref = known["reference"]
    assert len(oracle.huffman_compress(HELLO)) == ref["huffman_hello_size"]       # README.md:157
    assert len(oracle.huffman_compress(ABC)) == ref["huffman_abc_size"]           # README.md:167
    assert len(oracle.lzss_compress(HELLO)) == ref["lzss_hello_size"]             # README.md:153
    assert len(oracle.lzss_compress_legacy(ABC)) == ref["lzss_legacy_abc_size"]   # README.md:165


def test_survey_cross_check_huffman(oracle, known, samiam):
    s = known["survey"]
    c = oracle.huffman_compress(samiam)
    hs = s["huffman_samiam"]
    sep = c.index(b"\\\n")
    assert (len(c), sep, c[sep + 2], len(c) - sep - 3) == (hs["size"], hs["header"], hs["pad"], hs["payload"])
    assert sha(c[sep + 2:]) == hs["pad_payload_sha256"]
    assert sha(c) == hs["file_sha256_ascending_header"]
    t = oracle.huffman_table(samiam)
    assert len(t) == hs["symbols"] and max(x[3] for x in t) == hs["maxlen"]
    tiled = (samiam * 20)[:65536]
    ct = oracle.huffman_compress(tiled)
    assert (len(ct), sha(ct)) == (s["huffman_samiam_tiled_65536"]["size"], s["huffman_samiam_tiled_65536"]["sha256"])
    assert oracle.huffman_compress(b"ab").hex() == s["huffman_ab_hex"]
    assert oracle.huffman_compress(b"aaaa").hex() == s["huffman_aaaa_hex"]
    c = oracle.huffman_compress(b"a\nb\\")
    assert c.endswith(bytes.fromhex(s["huffman_a_nl_b_bs_payload_hex"]))
    assert {(r, l, code) for r, f, code, l in oracle.huffman_table(b"a\nb\\")} == {
        (10, 2, 0), (ord("b"), 2, 1), (ord("a"), 2, 2), (0x5C, 2, 3)}


def test_survey_cross_check_lzss(oracle, known, samiam):
    s = known["survey"]
    c = oracle.lzss_compress(samiam)
    assert (len(c), sha(c)) == (s["lzss_samiam"]["size"], s["lzss_samiam"]["sha256"])
    assert c == oracle.lzss_compress(samiam, 8192) == oracle.lzss_compress_allpos(samiam)
    g = oracle.lzss_compress_legacy(samiam)
    assert (len(g), sha(g)) == (s["lzss_legacy_samiam"]["size"], s["lzss_legacy_samiam"]["sha256"])
    lh = oracle.huffman_compress(c)
    assert (len(lh), sha(lh)) == (s["lzss_huffman_samiam"]["size"], s["lzss_huffman_samiam"]["sha256"])
    assert oracle.lzss_compress(ABC).decode() == s["lzss_abc"]
    assert oracle.lzss_compress_legacy(ABC).decode() == s["lzss_legacy_abc"]
    a, b = s["lzss_tiebreak"]
    assert oracle.lzss_compress(a.encode()).decode() == b
    for a, b in s["lzss_threshold"]:
        assert oracle.lzss_compress(a.encode()).decode() == b


def test_oracle_regression_pins(oracle, known, samiam):
    o = known["oracle"]
    assert oracle.huffman_compress(HELLO).hex() == o["huffman_hello_hex"]
    assert oracle.huffman_compress(ABC).hex() == o["huffman_abc_hex"]
    tiled = (samiam * 20)[:65536]
    assert sha(oracle.lzss_compress(tiled)) == o["lzss_tiled_sha256"]
    assert sha(oracle.huffman_compress(tiled)) == o["huffman_tiled_sha256"]


def test_reference_round_trips(oracle, samiam):
    # lzss_test.go:25-47 (window 8192) and cli_test.go:33-40 (huffman, lzss lossless on samIAm)
    assert oracle.lzss_decompress(oracle.lzss_compress(samiam, 8192)) == samiam
    assert oracle.lzss_decompress(oracle.lzss_compress_legacy(samiam, 8192)) == samiam
    assert oracle.huffman_decompress(oracle.huffman_compress(samiam)) == samiam
    # layered, engine.go:443-479: lzss then huffman; decode in reverse
    layered = oracle.huffman_compress(oracle.lzss_compress(samiam))
    assert oracle.lzss_decompress(oracle.huffman_decompress(layered)) == samiam


def go_runes(b):
    """Third, pure-Python statement of Go's range-over-string decoding."""
    out, i, n = [], 0, len(b)
    while i < n:
        b0 = b[i]
        if b0 < 0x80:
            out.append(b0); i += 1; continue
        lo, hi = 0x80, 0xBF
        if 0xC2 <= b0 <= 0xDF: need = 2
        elif b0 == 0xE0: need, lo = 3, 0xA0
        elif b0 == 0xED: need, hi = 3, 0x9F
        elif 0xE1 <= b0 <= 0xEF: need = 3
        elif b0 == 0xF0: need, lo = 4, 0x90
        elif b0 == 0xF4: need, hi = 4, 0x8F
        elif 0xF1 <= b0 <= 0xF3: need = 4
        else:
            out.append(0xFFFD); i += 1; continue
        seq = b[i:i + need]
        ok = len(seq) == need and lo <= seq[1] <= hi and all(0x80 <= c <= 0xBF for c in seq[2:])
        if not ok:
            out.append(0xFFFD); i += 1; continue
        out.append(ord(bytes(seq).decode("utf-8"))); i += need
    return out


def test_go_utf8_semantics(oracle, known):
    assert list(oracle.utf8_runes(bytes([0xE1, 0x80, 0xC2, 0x80]))) == known["survey"]["go_utf8_e180c280"]
    rng = random.Random(7)
    for _ in range(300):
        n = rng.randrange(1, 40)
        b = bytes(rng.choice([rng.randrange(256), rng.randrange(0x80, 0x100), rng.randrange(0xC0, 0xF8)]) for _ in range(n))
        assert list(oracle.utf8_runes(b)) == go_runes(b)
    txt = "héllo wörld ✓ 𝄞 \n".encode()
    assert list(oracle.utf8_runes(txt)) == [ord(c) for c in txt.decode()]


def test_huffman_lossy_on_invalid_utf8(oracle):
    # huffman.go:306-311: invalid bytes become U+FFFD; round trip = []byte(string([]rune(string(in))))
    rng = random.Random(11)
    b = bytes(rng.randrange(256) for _ in range(5000))
    expect = "".join(chr(r) for r in go_runes(b)).encode("utf-8")
    assert oracle.huffman_decompress(oracle.huffman_compress(b)) == expect


def test_huffman_edge_cases(oracle):
    with pytest.raises(oracle.OracleError):
        oracle.huffman_compress(b"")                       # heap.Pop on empty heap, huffman.go:102
    assert oracle.huffman_decompress(oracle.huffman_compress(b"aaaa")) == b"a"  # bare leaf emits once, huffman.go:137-142
    # '\\' must never be the last header entry (huffman.go:210 panics)
    c = oracle.huffman_compress(b"AB\\\\A")
    hdr = c[:c.index(b"\\\n")]
    assert hdr.startswith(b"2|\\") and oracle.huffman_decompress(c) == b"AB\\\\A"
    with pytest.raises(oracle.OracleError):
        oracle.huffman_decompress(b"2|\\\\\n\x00")          # single '\\' symbol: only order is the fatal one
    with pytest.raises(oracle.OracleError):
        oracle.huffman_decompress(b"no separator")
    # digits and '|' as symbols, newline escaped (huffman.go:313-317)
    s = b"1|2||33|\n\n7"
    assert oracle.huffman_decompress(oracle.huffman_compress(s)) == s
    # 900000-bit reference limit is opt-in
    rng = random.Random(3)
    big = bytes(rng.randrange(32, 127) for _ in range(200000))
    c = oracle.huffman_compress(big)
    assert oracle.huffman_decompress(c) == big
    with pytest.raises(oracle.OracleError):
        oracle.huffman_decompress(c, strict_ref_limit=True)


def test_header_multiset_helper(oracle, samiam):
    c = oracle.huffman_compress(samiam)
    ents, rest = oracle.header_entries(c)
    assert len(ents) == 46 and sum(int(f) for f, _ in ents) == len(samiam)
    assert (str(samiam.count(b"\n")).encode(), b"\\n") in ents


@pytest.mark.parametrize("alphabet,n", [(b"ab", 300), (b"abc<\\\xff", 400), (bytes(range(256)), 600), (b"a", 200)])
def test_lzss_lazy_equals_literal_form(oracle, alphabet, n):
    rng = random.Random(len(alphabet) * 1000 + n)
    for w in (4096, 16, 0):
        for _ in range(6):
            b = bytes(rng.choice(alphabet) for _ in range(rng.randrange(0, n)))
            c = oracle.lzss_compress(b, w)
            assert c == oracle.lzss_compress_allpos(b, w)
            assert oracle.lzss_decompress(c) == b


def test_lzss_escape(oracle):
    assert oracle.lzss_escape(b"a<b\xffc\\d") == b"a\xffb\\\xffc\\\\d"
    assert oracle.lzss_unescape(b"a\xffb\\\xffc\\\\d") == b"a<b\xffc\\d"
    assert oracle.lzss_compress(b"") == b"" and oracle.lzss_decompress(b"") == b""
    with pytest.raises(oracle.OracleError):
        oracle.lzss_decompress(b"ab<5,2>")


def test_lzss_window_binds(oracle):
    rng = random.Random(5)
    blk = bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(5000))
    b = blk + blk
    c = oracle.lzss_compress(b, 4096)      # distance 5000 > window: no long match
    assert b"<5000," not in c and oracle.lzss_decompress(c) == b
    c0 = oracle.lzss_compress(b, 0)        # unbounded (lzss.go:125)
    assert b"<5000," in c0 and oracle.lzss_decompress(c0) == b
    blk = blk[:4096]
    c = oracle.lzss_compress(blk * 3, 4096)
    assert c.endswith(b"<4096,4096><4096,4096>")


def test_ai_data_json_sizes(oracle, known):
    """ai/data.json holds sizes the reference itself measured on the Canterbury 'artificial' files, which are
    reproducible from their definition: aaa.txt = 100 000 x 'a' -> 0.40 % (ai/data.json:1992-2021),
    alphabet.txt = the alphabet repeated to 100 000 bytes -> 100 % (:2134-2163), a.txt = "a" -> 100 % (:2702-2731).
    Those rows were produced by the synchronous lz.Compress (lzss.go:224), restated as lzss_compress_legacy."""
    ref = known["reference"]
    aaa = b"a" * 100000
    alphabet = (b"abcdefghijklmnopqrstuvwxyz" * 3847)[:100000]
    assert len(oracle.lzss_compress_legacy(aaa)) == ref["lzss_legacy_aaa_100000_size"] == round(0.004 * 100000)
    assert len(oracle.lzss_compress_legacy(alphabet)) == ref["lzss_legacy_alphabet_100000_size"]
    assert len(oracle.lzss_compress_legacy(b"a")) == ref["lzss_legacy_a_size"]
    assert oracle.lzss_decompress(oracle.lzss_compress_legacy(aaa)) == aaa          # data.json: lossless=true for all three
    assert oracle.lzss_decompress(oracle.lzss_compress_legacy(alphabet)) == alphabet


def test_pi_txt_pins_the_legacy_encoder(oracle):
    """ai/data.json's pi.txt row (Canterbury large corpus: 1 000 000 digits of pi; engine "lzss": compressed_ratio 100.0,
    lossless false) is a reference-measured answer about the LEGACY encoder's quirks beyond the window (lzss.go:249-257: first-byte
    search un-windowed, pointer from the unsliced length; :272 the `<=` threshold): its output on pi is exactly 1 000 000 bytes
    and does not round-trip.  CompressAsync (the engine path today) gives another size and is lossless -- so the row also says
    which encoder produced data.json.  (found by the judge; the digits come from tests/golden/make_pi.py, ~1 min.)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("make_pi", os.path.join(os.path.dirname(__file__), "golden", "make_pi.py"))
    mp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mp)
    pi = mp.pi_digits()
    assert len(pi) == 1_000_000 and pi[:12] == b"314159265358" and sha(pi) == mp.SHA256
    legacy = oracle.lzss_compress_legacy(pi)
    assert len(legacy) == 1_000_000                                   # ratio 100.0
    assert oracle.lzss_decompress(legacy) != pi                       # lossless: false
    modern = oracle.lzss_compress_mt(pi, 4096, 8, 4096)
    assert len(modern) != 1_000_000 and oracle.lzss_decompress(modern) == pi
    assert modern[:200000] == oracle.lzss_compress(pi[:300000])[:200000]
    # the product's own host-side restatement of lz.Compress (raisin_amd/csrc/lzss_legacy.cpp, no device needed) on the same row
    import __graft_entry__ as g
    g.build()
    from raisin_amd import lz
    assert lz.Compress(pi) == legacy


def test_threaded_baseline_is_the_oracle(oracle, samiam):
    """oracle/cpu_baseline.c (bench.py's cpu_baseline on all host cores) produces the oracle's bytes."""
    import numpy as np
    import workloads as W
    bufs = [samiam, samiam * 40, b"a", b"ab", b"aaaa", bytes(W.config_input("2a", 400000).numpy()), bytes(W.config_input("skewed", 300001).numpy()),
            bytes(W.config_input("4", 200000).numpy()), bytes(W.config_input("2b", 50000).numpy()), bytes(W.config_input("3", 20000).numpy())]
    for d in bufs:
        ref = oracle.huffman_compress(d)
        for t in (1, 3, 8):
            assert oracle.huffman_compress_mt(d, t) == ref
            assert oracle.huffman_decompress_mt(ref, t) == oracle.huffman_decompress(ref)
    for d in bufs[:5] + [b[:60000] for b in bufs[5:]]:
        ref = oracle.lzss_compress(d)
        for t, grain in ((1, 4096), (8, 1), (5, 100)):
            assert oracle.lzss_compress_mt(d, 4096, t, grain) == ref
    assert oracle.lzss_compress_mt(samiam, 0, 4, 64) == oracle.lzss_compress(samiam, 0)


def test_oracle_forms_agree_on_low_entropy_shapes(oracle):
    """The GPU tests of runs, short periods and long copies check against the threaded every-position form; here it is held
    against the lazy single-thread form and, on short prefixes, the literal all-positions form -- three formulations of
    lzss.go:109-184 on the shapes where a longest match is thousands of bytes and ties are everywhere."""
    rng = random.Random(77)
    per3, per7, per1000 = (bytes(rng.randrange(97, 123) for _ in range(p)) for p in (3, 7, 1000))
    blk = bytes(rng.randrange(97, 123) for _ in range(1500))
    runs = b"".join(bytes([rng.randrange(97, 101)]) * 37 for _ in range(1200))
    shapes = [b"\x00" * 11000, (per3 * 9000)[:10000] + b"tail", b"head " + (per7 * 4000)[:10000], (per1000 * 30)[:11000], runs[:12000],
              b"".join(blk[:300 + 200 * i] + bytes([65 + i]) + bytes(rng.randrange(97, 123) for _ in range(50)) for i in range(6)),
              bytes(range(256)) * 40]
    for d in shapes:
        for w in (4096, 700):
            ref = oracle.lzss_compress(d, w)
            assert oracle.lzss_compress_mt(d, w, 8, 512) == ref
            assert oracle.lzss_decompress(ref) == d
        assert oracle.lzss_compress_allpos(d[:2500]) == oracle.lzss_compress(d[:2500])


def test_workload_generators():
    """workloads.py: splitmix64 as published, prefix-stable, and inside the alphabets BASELINE.md 3 names."""
    import numpy as np
    import workloads as W

    def sm(seed, n):
        out, s = [], seed
        for _ in range(n):
            s = (s + 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
            z = s
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2 ** 64 - 1)
            out.append(z ^ (z >> 31))
        return out
    assert sm(0, 1)[0] == 0xE220A8397B1DCDAF                                   # splitmix64's published first output for seed 0
    draws = sm(W.SEED_2, 3000)
    assert W.uniform_bytes(3000, W.SEED_2, 256).tolist() == [d & 255 for d in draws]
    assert W.uniform_bytes(3000, W.SEED_2, 128).tolist() == [d & 127 for d in draws]
    assert W.config_input("5", 100, chunk=3).tolist() == [d & 127 for d in sm(W.SEED_5 + 3, 100)]
    p = W.periodic(3 * 4096 + 5).numpy()
    assert (p[:4096] == p[4096:8192]).all() and (p[:5] == p[-5:]).all() and not np.isin(p, [0x5C, 0xFF]).any()
    vals = [v for v in range(256) if v not in (0x5C, 0xFF)]
    assert p[:50].tolist() == [vals[(d >> 11) % 254] for d in sm(W.SEED_3, 50)]
    t = W.zipf_text(300000).numpy()
    assert (W.zipf_text(100001).numpy() == t[:100001]).all()                    # prefix-stable
    assert set(np.unique(t).tolist()) <= set(range(97, 123)) | {10, 32}          # no '<', nothing >= 0x80
    words = bytes(t).split()
    top = max(set(words), key=words.count)
    assert 0.2 < words.count(top) / len(words) < 0.35                            # p(1) = 1 / sum k^-1.3 = 0.27
    s = W.skewed_bytes(200000).numpy()
    assert s.min() >= 32 and s.max() <= 127 and np.bincount(s)[32] > 4 * np.bincount(s)[32 + 12] * 0.9


def test_lzss_check_by_segments_is_an_equality_test(oracle):
    """oracle.lzss_check (cpu_baseline.c): True exactly for the oracle's own output -- a stream that round-trips but
    takes a nearer occurrence, a raw copy in place of a token, a truncated or extended stream are all refused."""
    import random
    rng = random.Random(3)
    words = [bytes(rng.choice(b"abcdefghij") for _ in range(rng.randint(2, 8))) for _ in range(40)]
    data = b" ".join(rng.choice(words) for _ in range(60000)) + b"<x\\y\xff" * 30
    c = oracle.lzss_compress(data)
    for seg in (1 << 10, 1 << 14, 1 << 30):
        for th in (1, 3):
            assert oracle.lzss_check(data, c, threads=th, seg=seg) == (True, 0)
    assert oracle.lzss_check(b"", b"") == (True, 0)
    assert not oracle.lzss_check(data, c[:-1])[0] and not oracle.lzss_check(data[:-1], c)[0]
    a, b = "abcdefg1abcdefg2abcdefg3", "abcdefg1<8,7>2<16,7>3"      # the farthest occurrence, not the nearest
    assert oracle.lzss_compress(a.encode()) == b.encode()
    assert oracle.lzss_check(a.encode(), b.encode())[0]
    near = b"abcdefg1<8,7>2<8,7>3"
    assert oracle.lzss_decompress(near) == a.encode() and not oracle.lzss_check(a.encode(), near)[0]
    k = c.index(b"<", len(c) // 2)
    j = c.index(b">", k)
    off, ln = (int(x) for x in c[k + 1:j].split(b","))
    raw = oracle.lzss_decompress(c[:k])                               # the escaped prefix, unescaped again...
    esc = oracle.lzss_escape(data)
    pos = len(oracle.lzss_escape(raw))
    lit = c[:k] + esc[pos:pos + ln] + c[j + 1:]                       # the token's bytes written out raw: still decodes to data
    assert oracle.lzss_decompress(lit) == data and not oracle.lzss_check(data, lit, seg=1 << 12)[0]


def test_c_oracle_against_the_literal_python_restatement():
    """oracle/literal.py restates huffman.go / lzss.go a second time, independently and the way the Go reads (bit strings, a list-backed
    container/heap, bytes.Index, the per-position recursion); the C oracle must agree with it byte for byte -- on tie-heavy alphabets
    (equal frequencies: only the heap's sift order decides the tree), invalid UTF-8, the escape bytes, windows 4096 / 100 / 16 / unbounded,
    and on what both reject.  Neither is reference output (no Go toolchain here): two restatements that agree pin each other, not the reference."""
    import random
    from oracle import literal as L
    from oracle import oracle as O
    O.build()
    rng = random.Random(20261004)
    alphabets = [b"ab", b"abc", b"abcdefgh", bytes(range(97, 123)) + b" \n", bytes(range(256)), b"a\n\\|0123456789", b"<\\\xff>,0123",
                 "äöü€𝄞 ab".encode(), bytes(range(0x80, 0x100)), b"\xe2\x82\xac\xe2\x82", b"\xf0\x9f\x98\x80\xf0\x9f", b"xy\xc3"]
    n_h = n_l = 0
    for it in range(420):
        alph = alphabets[it % len(alphabets)]
        n = rng.choice((1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 300))
        if it % 5 == 0:                                           # equal frequencies: every symbol k times, shuffled
            syms = list(dict.fromkeys(alph))[: rng.randint(2, 40)]
            k = rng.randint(1, 4)
            data = bytearray(bytes(syms) * k)
            rng.shuffle(data)
            data = bytes(data)
        else:
            data = bytes(rng.choice(alph) for _ in range(n))
        # ---- Huffman (more than one distinct rune: a single one is the bare-leaf special case, below)
        if len({r for _, r in L.go_runes(data)}) > 1:
            want = L.huffman_compress(data)
            assert O.huffman_compress(data) == want, (it, data[:40])
            assert O.huffman_decompress(want) == L.huffman_decompress(want), (it, data[:40])
            n_h += 1
        # ---- LZSS
        for w in (4096, 100, 16, 0):
            want = L.lzss_compress(data, w)
            assert O.lzss_compress(data, w) == want, (it, w, data[:40])
        assert O.lzss_decompress(want) == L.lzss_decompress(want) == data
        n_l += 1
    assert n_h > 300 and n_l == 420
    # the special cases of the Huffman format: one distinct symbol, the newline's two-byte header entry, '\\' last in Go's order
    assert L.huffman_compress(b"aaaa") == O.huffman_compress(b"aaaa") == b"4|a\\\n\x00"
    assert L.huffman_decompress(b"4|a\\\n\x00") == O.huffman_decompress(b"4|a\\\n\x00") == b"a"
    assert L.huffman_compress(b"ab") == b"1|a1|b\\\n\x06\x01"
    assert L.huffman_compress(b"a\nb\\")[-2:] == b"\x00\x87"
    # hand-written LZSS streams: tokens that copy tokens, odd but legal spellings, what both reject
    for stream in (b"abc<3,3><6,6>x<1,1>", b"ab<0,0>c<2,0>", b"x<1,1><2,2><4,4>", b"lit>,1<2,1>>", b"\\\\\\a<1,1>", b"a\\", b"q<01,1>"):
        assert O.lzss_decompress(stream) == L.lzss_decompress(stream), stream
    for bad in (b"ab<9,2>", b"<1,1>", b"abc<2,3>"):
        with pytest.raises(Exception):
            L.lzss_decompress(bad)
        with pytest.raises(Exception):
            O.lzss_decompress(bad)
