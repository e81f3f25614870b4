"""GPU parity: librsn LZSS encode/decode (through the C ABI) vs the CPU oracle, bit-exact."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lz():
    from raisin_amd import lz
    return lz


def rnd(seed, n, alphabet):
    rng = np.random.default_rng(seed)
    a = np.frombuffer(bytes(alphabet), dtype=np.uint8)
    return a[rng.integers(0, len(a), size=n)].tobytes()


def text(seed, n):
    rng = random.Random(seed)
    words = [b"the", b"quick", b"brown", b"fox", b"jumps", b"over", b"lazy", b"dog", b"compression", b"raisin",
             b"huffman", b"lzss", b"window", b"buffer", b"a", b"I", b"<tag>", b"back\\slash", b"\xff\xfe"]
    out = bytearray()
    while len(out) < n:
        out += rng.choice(words) + rng.choice([b" ", b" ", b"\n", b", "])
    return bytes(out[:n])


def test_fixtures_and_known_answers(lz, oracle, samiam, known):
    s = known["survey"]
    assert lz.CompressAsync(b"Hello world!\n") == b"Hello world!\n"
    assert lz.CompressAsync(b"abc" * 8 + b"\n").decode() == s["lzss_abc"]
    a, b = s["lzss_tiebreak"]
    assert lz.CompressAsync(a.encode()).decode() == b          # leftmost (farthest) occurrence wins
    for a, b in s["lzss_threshold"]:
        assert lz.CompressAsync(a.encode()).decode() == b      # token only if strictly shorter (lzss.go:143)
    c = lz.CompressAsync(samiam, False, 8192)                 # lzss_test.go:37-47
    assert c == oracle.lzss_compress(samiam, 8192)
    assert lz.Decompress(c, False) == samiam
    assert lz.CompressAsync(samiam) == oracle.lzss_compress(samiam, 4096)
    assert lz.CompressAsync(b"") == b"" and lz.Decompress(b"") == b""


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 129, 4095, 4096, 4097, 8191, 8193, 16383, 16385, 40000])
def test_sizes_small_alphabet(lz, oracle, n):
    data = rnd(n, n, b"ab")
    c = lz.CompressAsync(data)
    assert c == oracle.lzss_compress(data)
    assert lz.Decompress(c) == data


@pytest.mark.parametrize("alphabet", [b"a", b"abc<\\\xff", bytes(range(256)), b"\\", b"<", b"\xff\\"])
def test_alphabets_with_escapes(lz, oracle, alphabet):
    for n in (300, 5000, 20011):
        data = rnd(len(alphabet) * 7 + n, n, alphabet)
        c = lz.CompressAsync(data)
        assert c == oracle.lzss_compress(data)
        assert lz.Decompress(c) == oracle.lzss_decompress(c) == data


@pytest.mark.parametrize("w", [1, 5, 16, 63, 64, 65, 255, 1000, 4096, 8192])
def test_windows(lz, oracle, w):
    data = text(w, 30000)
    c = lz.CompressAsync(data, False, w)
    assert c == oracle.lzss_compress(data, w)
    assert lz.Decompress(c) == data


@pytest.mark.parametrize("w", [0, 8193, 16384, 100000])
def test_windows_above_8192_and_unbounded(lz, oracle, w):
    """L2: NewWriterLevel(w, level) takes any level >= 0 and CompressAsync treats <= 0 as "the whole prefix"
    (lzss.go:42-51,123-127): exact at any window, through the large-window path (lzss_big.hip)."""
    data = text(w + 5, 320000) + long_copies(w + 1, 40000)
    c = lz.CompressAsync(data, False, w)
    assert c == oracle.lzss_compress(data, w)
    assert lz.Decompress(c) == data
    if w in (0, 100000):                                      # distances and lengths beyond 16 bits
        far = rnd(3, 70000, bytes(range(97, 123))) 
        data = far + rnd(4, 150000, b"xyz") + far
        c = lz.CompressAsync(data, False, w)
        assert c == oracle.lzss_compress(data, w) and lz.Decompress(c) == data
        assert (b"<220000,70000>" in c) == (w == 0)               # only the unbounded window reaches 220000 back


def test_large_window_refuses_pathological_runs_cleanly(lz):
    from raisin_amd import RsnError
    with pytest.raises(RsnError) as e:
        lz.CompressAsync(b"a" * 3000000, False, 0)
    assert e.value.code == -6 and "work budget" in str(e.value)
    assert lz.Decompress(lz.CompressAsync(b"a" * 300000)) == b"a" * 300000     # the default window has no such limit


def test_unbounded_window_small(lz, oracle):
    data = text(3, 6000)
    assert lz.CompressAsync(data, False, 0) == oracle.lzss_compress(data, 0)   # lzss.go:125: <=0 means no limit
    assert lz.CompressAsync(data, False, 100000) == oracle.lzss_compress(data, 100000)


def test_text_and_period(lz, oracle):
    data = text(11, 300000)
    c = lz.CompressAsync(data)
    assert c == oracle.lzss_compress(data)
    assert lz.Decompress(c) == data
    # config-3 shape: one 4096-byte block (no 5C / FF) repeated
    rng = np.random.default_rng(0x5EED0003)
    vals = np.array([v for v in range(256) if v not in (0x5C, 0xFF)], dtype=np.uint8)
    blk = vals[rng.integers(0, len(vals), size=4096)].tobytes()
    data = blk * 40
    c = lz.CompressAsync(data)
    assert c == oracle.lzss_compress(data)
    assert c.endswith(b"<4096,4096>" * 30)
    assert lz.Decompress(c) == data


def test_long_runs(lz, oracle):
    for data in (b"a" * 100000, b"ab" * 30000, b"\\" * 20001, (b"x" * 5000 + b"y") * 8):
        c = lz.CompressAsync(data)
        assert c == oracle.lzss_compress(data)
        assert lz.Decompress(c) == data


@pytest.mark.parametrize("n", [4095, 4096, 4097, 8191, 8192, 8193, 12289, 16383, 16384, 16385, 20481, 32767, 32769, 65537])
def test_sizes_text_tile_edges(lz, oracle, n):
    """Text keeps the bucket search on its own path (no hand-back): stream ends at, before and
    after every tile (4096) and strip (16384) edge; windows that do and do not reach back a tile."""
    data = text(n, n)
    for w in (4096, 1000):
        c = lz.CompressAsync(data, False, w)
        assert c == oracle.lzss_compress(data, w), (n, w)
    assert lz.Decompress(c) == data


@pytest.fixture(params=["groups by size", "groups of 128"])
def group_mode(request):
    """The decoder's chain groups: 4 tiles for streams up to 2048 tiles, n_tiles / 512 beyond, 128 from 1 GiB
    (lzss_decode.hip group_tiles); RSN_LZSS_DEC_GROUP128 fixes them at 128 whatever the size."""
    if request.param == "groups of 128":
        os.environ["RSN_LZSS_DEC_GROUP128"] = "1"
        yield request.param
        del os.environ["RSN_LZSS_DEC_GROUP128"]
    else:
        yield request.param


def test_decode_group_sizes_of_mid_streams(lz):
    """2561 tiles (40 MiB, groups of 5) and 8195 tiles (128 MiB, groups of 16): copies that chain across every tile
    and group, text between them; device-resident round trip."""
    import torch
    rng = np.random.default_rng(5)
    vals = np.array([v for v in range(256) if v not in (0x5C, 0xFF, 0x3C)], dtype=np.uint8)
    blk = vals[rng.integers(0, len(vals), size=4096)].tobytes()
    unit = text(9, 300000) + blk * 40 + text(10, 100000) + blk * 3
    for tiles in (2561, 8195):
        n = tiles * 16384 - 7
        src = torch.frombuffer(bytearray((unit * (n // len(unit) + 1))[:n]), dtype=torch.uint8).cuda()
        c = lz.compress_tensor(src)
        assert torch.equal(lz.decompress_tensor(c), src), tiles


@pytest.mark.parametrize("tiles", [1, 2, 3, 4, 5, 127, 128, 129, 256, 257])
def test_decode_tile_and_group_edges(lz, oracle, tiles, group_mode):
    """Escaped-stream lengths at, before and after the 16 KiB resolve tiles and the chain groups (of 4 tiles at
    these sizes, and of 128), on data whose copies chain across every tile (period) and on text."""
    rng = np.random.default_rng(tiles)
    vals = np.array([v for v in range(256) if v not in (0x5C, 0xFF, 0x3C)], dtype=np.uint8)
    blk = vals[rng.integers(0, len(vals), size=4096)].tobytes()
    for delta in (-1, 0, 1):
        n = tiles * 16384 + delta
        per = (blk * (n // 4096 + 1))[:n]
        c = lz.CompressAsync(per)
        assert lz.Decompress(c) == per, (tiles, delta)
    n = tiles * 16384 + 5
    t = text(tiles, min(n, 600000))
    mixed = (t + per)[:n] if len(t) < n else t[:n]
    c = lz.CompressAsync(mixed)
    assert c == oracle.lzss_compress(mixed)
    assert lz.Decompress(c) == oracle.lzss_decompress(c) == mixed


def test_bucket_search_hands_strips_back(lz, oracle):
    """The bigram-bucket search gives a strip to the diagonal sweep when a common prefix reaches 256
    bytes or a bucket walk runs too long; both kinds of strip next to ordinary ones must still
    produce the oracle's bytes."""
    rng = np.random.default_rng(77)
    block = rng.integers(0, 256, size=1500, dtype=np.uint8).tobytes()
    repeats = b"".join(block + rng.integers(0, 256, size=int(rng.integers(1, 400)), dtype=np.uint8).tobytes() for _ in range(24))
    data = text(31, 40000) + rnd(5, 40000, b"ab") + repeats + rnd(6, 40000, bytes(range(256))) + rnd(7, 20000, b"abc") + text(32, 30000)
    c = lz.CompressAsync(data)
    assert c == oracle.lzss_compress(data)
    assert lz.Decompress(c) == data
    for w in (700, 2048):
        assert lz.CompressAsync(data[:120000], False, w) == oracle.lzss_compress(data[:120000], w)


def test_long_matches_followed_through_memory(lz, oracle):
    """Common prefixes of 256 bytes and more -- beyond what the chain walk's LDS stage holds.  Up to eight such candidates
    per position are followed through memory to their true end (min of common prefix, distance, bytes left); with more
    of them (runs, short periods) the farthest decides if it matches over its whole distance, else the strip goes to
    the sweep.  Every case against the oracle: single copies, several earlier copies of different lengths (longest wins,
    then farthest), more than eight copies, runs at the start / inside / at the end of a stream, short periods inside
    text, a long match cut by the end of the stream, windows smaller than the copies' distances."""
    rng = np.random.default_rng(123)

    def noise(k):
        return rng.integers(97, 123, size=k, dtype=np.uint8).tobytes()

    blk = noise(2000)
    cases = {
        "one copy": text(1, 9000) + blk + text(2, 1500) + blk[:1000] + b"!" + text(3, 9000),
        "copies of different lengths": text(4, 5000) + blk[:300] + b"1" + noise(200) + blk[:700] + b"2" + noise(150) + blk[:1900] + b"3" + noise(100)
                                       + blk[:1200] + b"4" + text(5, 6000),
        "equal lengths, farthest wins": text(6, 3000) + blk[:600] + b"A" + noise(50) + blk[:600] + b"B" + noise(70) + blk[:600] + b"C" + noise(20) + blk[:600] + b"D" + text(7, 3000),
        "twelve copies": text(8, 2000) + b"".join(blk[:280] + bytes([65 + i]) + noise(10 + 3 * i) for i in range(12)) + blk[:280] + b"z" + text(9, 4000),
        "run at the start": b"\x00" * 70000 + text(10, 20000),
        "runs inside and at the end": text(11, 10000) + b"a" * 3000 + text(12, 300) + b"a" * 9000 + b"b" * 5000 + text(13, 8000) + b"q" * 12345,
        "short periods inside text": text(14, 6000) + b"ab" * 4000 + text(15, 500) + b"xyz" * 3000 + text(16, 700) + (noise(7) * 2000) + text(17, 9000) + (noise(1000) * 9),
        "cut by the end of the stream": text(18, 12000) + blk + text(19, 700) + blk[:900],
    }
    for name, data in cases.items():
        for w in (4096, 1000):
            c = lz.CompressAsync(data, False, w)
            assert c == oracle.lzss_compress_mt(data, w, oracle.host_cores(), 4096), (name, w)
        assert lz.Decompress(c) == data, name


def test_long_candidates_where_the_end_of_the_stream_binds(lz, oracle):
    """A stream that ends in a run of zeros, an earlier zero run of 250..256 bytes inside the window, and the chain landing
    250..256 bytes before the end: there the many candidates of the run are capped by the bytes LEFT, not by their distance, and
    a candidate farther back than the farthest "long" one (the start of the earlier, shorter run) matches all of them too --
    bytes.Index takes that leftmost occurrence (lzss.go:419).  ADVICE r2: the farthest-long-candidate rule used to overwrite it."""
    rng = np.random.default_rng(77)
    head = rng.integers(97, 123, size=3000, dtype=np.uint8).tobytes()
    for run in (250, 252, 255, 256):
        for k in (1, 2, 5):
            for r in (0, 1, 3, 6, 8):
                for gap in (b"q", b"q" + head[:700]):
                    data = head + b"q" + b"\x00" * run + gap + b"\x00" * (256 * k + r)
                    c = lz.CompressAsync(data)
                    assert c == oracle.lzss_compress(data), (run, k, r, len(gap))
    data = head + b"q" + b"\x00" * 256 + b"q" + b"\x00" * (256 * 40)
    assert lz.CompressAsync(data) == oracle.lzss_compress(data)
    assert lz.Decompress(lz.CompressAsync(data)) == data


def test_joints_by_the_hundred_are_mended_by_one_look(lz, oracle):
    """Runs of 37 equal bytes from four letters: steps of 40 to 100 positions leave a tile's 128-position warm-up chain two or
    three steps to merge with the true chain, and a quarter of the joints fail.  The tile before's exit is right all the
    same (its chain merged further in), so one second look from there mends them all: no general parse."""
    rng = np.random.default_rng(37)
    data = np.repeat(rng.integers(97, 101, size=(3 << 20) // 37 + 1, dtype=np.uint8), 37)[: 3 << 20].tobytes()
    c, p = _prof(lz, data)
    assert c == oracle.lzss_compress_mt(data, 4096, oracle.host_cores(), 4096)
    assert lz.Decompress(c) == data
    if _chain_mode() and "RSN_LZSS_NO_FUSED_PARSE" not in os.environ:
        assert p["lzss_match_chain"][0] >= 2 and "lzss_parse_mark" not in p, sorted(p)


@pytest.mark.parametrize("period", [3, 7, 258, 1000])
def test_whole_distance_stretches_placed_by_arithmetic(lz, oracle, period):
    """A stretch that repeats with a period which does not divide the window: every match there is the farthest multiple of
    the period inside the window, over that whole distance, every chain steps by it and keeps its phase -- the per-tile
    chains never join.  k_match_chain reports such tiles, k_stretch_pred places the true chain through the run of them,
    one more look walks it: the oracle's bytes without the general parse.  Text before, between and after; a second
    stretch with another period; a stretch that runs to the end of the stream."""
    rng = np.random.default_rng(period)
    per = rng.integers(97, 123, size=period, dtype=np.uint8).tobytes()
    per2 = rng.integers(65, 91, size=period + 2, dtype=np.uint8).tobytes()
    data = text(period, 20000) + (per * (700000 // period + 1))[:700000] + text(period + 1, 30000) + (per2 * (650000 // len(per2) + 1))[:650000]
    for tail in (text(period + 2, 9000), b""):
        d = data + tail
        c, p = _prof(lz, d)
        assert c == oracle.lzss_compress_mt(d, 4096, oracle.host_cores(), 4096)
        assert lz.Decompress(c) == d
        if _chain_mode() and not os.environ.get("RSN_LZSS_NO_FUSED_PARSE"):
            # (without the tail the last match runs to the very end of the stream and jumps over the final partial tile, which then
            #  holds no chain position at all: k_chain_verify accepts that)
            assert "lzss_chain_stretch" in p and "lzss_parse_mark" not in p, sorted(p)


def test_noise_sections_do_not_keep_the_looks_from_the_rest(lz, oracle):
    """More tiles give up as dense (two sections of noise) than a look takes: the stream ends in the general parse, but the
    joints and the whole-distance stretch of the other sections are still mended and placed first.  Oracle's bytes."""
    rng = np.random.default_rng(99)
    noise = rng.integers(0, 256, size=560000, dtype=np.uint8).tobytes()
    per = rng.integers(97, 123, size=7, dtype=np.uint8).tobytes()
    runs = np.repeat(rng.integers(97, 101, size=300000 // 37 + 1, dtype=np.uint8), 37)[:300000].tobytes()
    data = text(1, 30000) + noise[:280000] + (per * 90000)[:600000] + text(2, 20000) + noise[280000:] + runs + text(3, 10000)
    c, p = _prof(lz, data)
    assert c == oracle.lzss_compress_mt(data, 4096, oracle.host_cores(), 4096)
    assert lz.Decompress(c) == data
    if _chain_mode() and not os.environ.get("RSN_LZSS_NO_FUSED_PARSE"):
        assert "lzss_chain_stretch" in p and p["lzss_match_chain"][0] >= 2, sorted(p)


def test_match_table_against_oracle(lz, oracle):
    """Chain-independent check: the oracle's greedy parse only ever looks at chain positions."""
    data = text(21, 50000)
    c = lz.CompressAsync(data)
    assert c == oracle.lzss_compress_allpos(data)


def test_decode_errors(lz, oracle):
    from raisin_amd import RsnError
    for bad in (b"ab<5,2>", b"abcdef<3,4>", b"abc<1,", b"abc<x,1>", b"abc<1,1"):
        try:
            want = oracle.lzss_decompress(bad)
        except oracle.OracleError:
            want = None
        if want is None:
            with pytest.raises(RsnError):
                lz.Decompress(bad)
    with pytest.raises(ValueError):
        lz.NewWriterLevel(None, -1)                            # lzss.go:43-45


def test_decode_front_end_in_one_pass(lz, oracle, monkeypatch):
    """The decoder's counting pass also leaves, per 4 KB block, how far into the output the block must start for its pointers to be
    valid, and per 16-byte span what it produces; the tile kernels then run without a second parse of the tokens (k_lzd_tilemap).
    Checked here: a pointer before the start of the data deep inside a stream (the compare per block, read back only after the tile
    kernels have run), at a block edge and inside a tile; spans that produce 65535 bytes and more (the pass hands those streams
    to k_lzd_tiles); tile boundaries that fall inside literals, inside tokens and right after zero-output spans; and the
    same bytes from the decoder's second formulation (RSN_LZSS_DEC_JUMP: pointer jumping over the whole stream) on everything."""
    from raisin_amd import RsnError
    rng = np.random.default_rng(9)
    lit = rng.integers(97, 123, size=5000, dtype=np.uint8).tobytes()
    for bad in (lit + b"<6000,10>" + lit, lit[:4090] + b"<4091,7>", lit * 8 + b"<40001,9>" + lit, b"<1,1>" + lit):
        with pytest.raises(oracle.OracleError):
            oracle.lzss_decompress(bad)
        with pytest.raises(RsnError) as ei:
            lz.Decompress(bad)
        assert ei.value.code == -3
    ok = lit + b"<5000,10>" + lit[:4086] + b"<4096,4096>" * 40 + b"<7,7>" * 3000 + lit + b"<16384,16000>" * 9
    big = rng.integers(97, 123, size=80000, dtype=np.uint8).tobytes()
    streams = [ok, big + b"<80000,80000>" + b"<160000,70000>x", text(77, 300000) and oracle.lzss_compress(text(77, 300000)),
               oracle.lzss_compress(lit * 40), (lit * 4)[:16384 - 5] + b"<9,9>" + (lit * 4)[:16384 - 9] + b"<16384,16384>" * 5 + b"q"]
    want = [oracle.lzss_decompress(c) for c in streams]
    assert [lz.Decompress(c) for c in streams] == want
    import subprocess, sys, hashlib
    code = ("import sys, hashlib; sys.path.insert(0, %r)\nfrom raisin_amd import lz\n"
            "import pickle; streams = pickle.load(open(sys.argv[1], 'rb'))\n"
            "print(' '.join(hashlib.sha256(lz.Decompress(c)).hexdigest() for c in streams))\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import pickle, tempfile
    with tempfile.NamedTemporaryFile(suffix=".pkl") as f:
        pickle.dump(streams, f)
        f.flush()
        out = subprocess.run([sys.executable, "-c", code, f.name], capture_output=True, text=True, timeout=300, env=dict(os.environ, RSN_LZSS_DEC_JUMP="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[-len(streams):] == [hashlib.sha256(w).hexdigest() for w in want]


def test_decode_run_tiles(lz, oracle):
    """Output tiles that hold nothing but a few tokens are resolved as intervals by k_lzd_runs (one wavefront per tile, no descriptors):
    W-periodic data with tile-aligned and straddling tokens, periods that do not divide the tile, runs of one byte, run tiles next
    to ordinary ones, tiles with more tokens than a run tile takes (32) or more resolved runs (64), hand-written streams whose tokens
    copy pieces of several earlier tokens; and the same bytes without the path (RSN_LZSS_DEC_NO_RUNS) in a process of its own."""
    import hashlib
    import pickle
    import subprocess
    import sys
    import tempfile
    rng = np.random.default_rng(21)
    blk = rng.integers(97, 123, size=4096, dtype=np.uint8).tobytes()
    datas = [blk * 40, blk[:1000] * 100, blk[:4095] * 30, blk[:37] * 3000, b"\x00" * 200000, text(5, 40000) + blk * 20 + text(6, 30000) + blk[:777] * 90,
             (blk[:300] + b"Q") * 400, blk * 3 + blk[:2048] * 31]
    streams = [oracle.lzss_compress(d) for d in datas]
    lit = rng.integers(97, 123, size=20000, dtype=np.uint8).tobytes()
    # hand-written: tokens that copy across earlier tokens' seams, 40 short tokens in one tile, a token of the full tile
    streams += [lit + b"<4096,4096>" * 3 + b"<6000,4100>" + b"<100,100>" * 200 + b"<16384,16384>" * 4 + b"<3,3>" * 20000,
                lit + (b"<9,9>" + b"<500,400>") * 300 + b"<16000,16000>" * 3,
                lit[:16384] + b"<16384,8192>" + b"<8192,8192>" + b"<12288,12288>" + b"<4096,4096>" * 9 + b"x"]
    want = [oracle.lzss_decompress(c) for c in streams]
    for d, w in zip(datas, want):
        assert d == w
    assert [lz.Decompress(c) for c in streams] == want
    code = ("import sys, hashlib, pickle; sys.path.insert(0, %r)\nfrom raisin_amd import lz\n"
            "streams = pickle.load(open(sys.argv[1], 'rb'))\n"
            "print(' '.join(hashlib.sha256(lz.Decompress(c)).hexdigest() for c in streams))\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    with tempfile.NamedTemporaryFile(suffix=".pkl") as f:
        pickle.dump(streams, f)
        f.flush()
        out = subprocess.run([sys.executable, "-c", code, f.name], capture_output=True, text=True, timeout=300, env=dict(os.environ, RSN_LZSS_DEC_NO_RUNS="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[-len(streams):] == [hashlib.sha256(w).hexdigest() for w in want]


def test_decode_paths(lz, oracle):
    """The tile path (back-pointers <= 16384), its fallbacks, and streams only a foreign encoder writes."""
    data = text(41, 70000)
    far = oracle.lzss_compress(data, 0)                        # unbounded window: pointers beyond a tile -> whole-stream pointer jumping
    assert lz.Decompress(far) == oracle.lzss_decompress(far) == data
    mid = oracle.lzss_compress(data, 16384)                    # tail = a whole tile
    assert lz.Decompress(mid) == data
    zero = b"abcdefgh" * 40 + b"<8,0>" * 6000 + b"xyz" * 9000 + b"<27000,27>" + b"<3,3>"   # zero-length tokens stretch a tile's input
    assert lz.Decompress(zero) == oracle.lzss_decompress(zero)
    hand = b"0123456789" + b"<10,10>" * 5000 + b"<4,4>" + b"q" * 20000 + b"<20000,20000><7,0>!"   # deep chains across many tiles
    assert lz.Decompress(hand) == oracle.lzss_decompress(hand)
    rng = np.random.default_rng(8)
    blk = rng.integers(0, 250, size=5000, dtype=np.uint8).tobytes().replace(b"<", b"x").replace(b"\\", b"y")
    long_tok = blk + b"<5000,5000>" * 30 + b"<16000,16000>" * 3   # tokens that span tile boundaries
    assert lz.Decompress(long_tok) == oracle.lzss_decompress(long_tok)
    long_tok += b"<150000,100000>"                              # one spanning several tiles: general path
    assert lz.Decompress(long_tok) == oracle.lzss_decompress(long_tok)


def test_layered_lzss_then_huffman(lz, oracle, samiam, known):
    from raisin_amd import huffman
    layered = huffman.Compress(lz.CompressAsync(samiam))      # engine.go:443-452
    assert len(layered) == known["survey"]["lzss_huffman_samiam"]["size"]
    assert layered == oracle.huffman_compress(oracle.lzss_compress(samiam))
    assert lz.Decompress(huffman.Decompress(layered)) == samiam   # engine.go:454-479 (reverse order)


def test_device_resident_16MiB(lz, oracle):
    import torch
    data = text(5, 16 << 20)
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    c = lz.compress_tensor(src)
    d = lz.decompress_tensor(c)
    assert torch.equal(d, src)
    assert bytes(c.cpu().numpy()) == oracle.lzss_compress(data)


# ---- chain walk (k_match_chain): keys only where greedy chains land -------------------------
def long_copies(seed, n, seg=(90, 240)):
    """Matches of 90..240 bytes from up to 3000 back, one fresh byte between them: a chain step is
    about as long as the warm-up, so the warm-up chain often has NOT merged with the
    true chain when it reaches its tile and the parse lands on unevaluated positions."""
    rng = np.random.default_rng(seed)
    out = bytearray(rng.integers(97, 123, size=4096, dtype=np.uint8).tobytes())
    while len(out) < n:
        ln = int(rng.integers(seg[0], seg[1]))
        back = int(rng.integers(ln + 1, 3000))
        start = len(out) - back
        out += out[start:start + ln]
        out.append(int(rng.integers(65, 91)))
    return bytes(out[:n])


def _chain_mode():
    """False when an A/B switch replaces the chain walk (the suites are also run under those switches)."""
    import os
    return not os.environ.get("RSN_LZSS_ALLPOS")


def _prof(lz_mod, data, w=4096):
    from raisin_amd import _lib
    _lib.prof_enable(True)
    _lib.prof_reset()
    c = lz_mod.CompressAsync(data, False, w)
    p = _lib.prof_get()
    _lib.prof_enable(False)
    return c, p


def test_chain_redo_round(lz, oracle):
    """The true chain meets a position no speculative chain evaluated: those strips are searched at
    every position and the parse is repeated -- same bytes as the oracle.  A 200-periodic stream
    under a 300-byte window has L = 200 everywhere, so chains of different phase never merge.
    (r05: the walk places such a stream's chain by arithmetic even when it is only a few tiles long, so the round is reached through the
    general parse -- RSN_LZSS_NO_FUSED_PARSE=1, a process of its own: the switch is read once.)"""
    import hashlib
    import os
    import subprocess
    import sys
    blk = rnd(5, 200, bytes(range(97, 123)))
    cases = ((60000, 300), (100001, 250), (50000, 201))
    want = []
    for n, w in cases:
        data = (blk * (n // 200 + 1))[:n]
        c = lz.CompressAsync(data, False, w)
        assert c == oracle.lzss_compress(data, w)
        assert lz.Decompress(c) == data
        want.append(hashlib.sha256(c).hexdigest())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "from raisin_amd import lz, _lib\n"
            "blk = bytes.fromhex(%r)\n"
            "for n, w in %r:\n"
            "    data = (blk * (n // 200 + 1))[:n]\n"
            "    _lib.prof_enable(True); _lib.prof_reset()\n"
            "    c = lz.CompressAsync(data, False, w)\n"
            "    p = _lib.prof_get(); _lib.prof_enable(False)\n"
            "    print(hashlib.sha256(c).hexdigest(), p.get('lzss_parse_exit', (0, 0))[0])\n" % (root, blk.hex(), cases))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, RSN_LZSS_NO_FUSED_PARSE="1", RSN_LZSS_NO_PREFLAG="1"))   # (r06: without the second switch the listed tiles' strips are searched before the first parse)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [x.split() for x in out.stdout.strip().splitlines()[-3:]]
    assert [r[0] for r in rows] == want
    assert all(int(r[1]) > 1 for r in rows) or not _chain_mode(), "the redo round was not exercised: %r" % rows
    for seed in (1, 2):                      # long copies with natural re-synchronisation points
        data = long_copies(seed, 120000)
        assert lz.CompressAsync(data) == oracle.lzss_compress(data)


def test_chain_dense_and_mixed(lz, oracle):
    """Incompressible stretches are handed to the all-positions search strip by strip."""
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, size=70000, dtype=np.uint8).tobytes()
    mixed = text(21, 50000) + noise + text(22, 40000) + b"ab" * 9000 + noise[:33000] + long_copies(9, 30000)
    for data in (noise, mixed):
        c, p = _prof(lz, data)
        assert not _chain_mode() or p["lzss_match_chain"][0] >= 1
        assert c == oracle.lzss_compress(data)
        assert lz.Decompress(c) == data


def test_chain_tile_edges(lz, oracle):
    """Sizes around the 8192-position chain tile, its warm-up (128 positions; 256 earlier) and the 16384 strip."""
    base = text(31, 40000)
    for n in (8191, 8192, 8193, 8192 + 127, 8192 + 128, 8192 + 129, 8192 + 255, 8192 + 256, 8192 + 257, 16383, 16384, 16385, 24576 + 1, 32768):
        data = base[:n]
        assert lz.CompressAsync(data) == oracle.lzss_compress(data), n


def test_chain_period_inside_first_tile(lz, oracle):
    """A W-periodic stretch that starts inside a tile (config 3's first tile): answered position by
    position (nothing beats L = W at distance W) instead of going to the sweep."""
    blk = rnd(77, 4096, bytes(v for v in range(256) if v not in (0x5C, 0xFF, 0x3C)))
    data = blk * 9 + b"tail" + blk[:1000]
    c, p = _prof(lz, data)
    assert c == oracle.lzss_compress(data)
    assert lz.Decompress(c) == data
    for w in (1000, 4095):
        data = blk[:w] * 12
        assert lz.CompressAsync(data, False, w) == oracle.lzss_compress(data, w)


def test_chain_periodic_stretches_between_text(lz, oracle):
    """W-periodic tiles have no chain of their own (phases never merge in periodic data): their chain is placed by
    arithmetic from the last walked tile's exit (k_chain_periodic) and must join the walked tiles on both sides."""
    blk = rnd(78, 4096, bytes(v for v in range(256) if v not in (0x5C, 0xFF, 0x3C)))
    for w, per, reps in ((4096, 4096, 21), (4096, 4096, 5), (3000, 3000, 40), (1024, 1024, 70)):
        data = text(41, 30000) + blk[:per] * reps + text(42, 41000) + blk[:per] * (reps + 3) + b"end"
        c, p = _prof(lz, data, w)
        assert c == oracle.lzss_compress(data, w), (w, per, reps)
        assert lz.Decompress(c) == data


def test_chain_vs_allpos_switch(oracle):
    """RSN_LZSS_ALLPOS=1 (bucket search at every position) gives the same bytes, and so does the in-tile parse by
    one lane per tile (RSN_LZSS_TAIL_SERIAL: what streams of 256 MiB and more take; below, a block per tile does
    it) -- through the records k_match_chain leaves it and, RSN_LZSS_NO_CKEYS, through the key array -- and the general
    parse instead of the walk's own (RSN_LZSS_NO_FUSED_PARSE): separate processes, the switches are read once.  The input has text, long copies, runs and a short period in it (the walk's long-match
    paths: candidates followed through memory, the early end of a bucketful visit)."""
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from raisin_amd import lz\n"
            "import hashlib\n"
            "from tests.test_gpu_lzss import text, long_copies\n"
            "d = text(5, 150000) + long_copies(4, 60000) + b'ab' * 9000 + bytes(20000) + text(6, 30000) + b'xyz' * 7000 + text(7, 9000)\n"
            "print(hashlib.sha256(lz.CompressAsync(d)).hexdigest())\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    envs = ({}, {"RSN_LZSS_ALLPOS": "1"}, {"RSN_LZSS_TAIL_SERIAL": "1"}, {"RSN_LZSS_TAIL_SERIAL": "1", "RSN_LZSS_NO_CKEYS": "1"},
            {"RSN_LZSS_NO_FUSED_PARSE": "1"}, {"RSN_LZSS_ALLPOS": "1", "RSN_LZSS_NO_FUSED_PARSE": "1"})
    for env in envs:
        e = dict(os.environ); e.update(env)
        outs.append(subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, check=True).stdout.strip())
    import hashlib
    d = text(5, 150000) + long_copies(4, 60000) + b"ab" * 9000 + bytes(20000) + text(6, 30000) + b"xyz" * 7000 + text(7, 9000)
    want = hashlib.sha256(oracle.lzss_compress(d)).hexdigest()
    assert [o == want for o in outs] == [True] * len(envs), list(zip(envs, outs))


@pytest.mark.parametrize("shift", [0, 1, 2, 3, 4, 5])
def test_escape_blocks_plain_and_mixed(lz, oracle, shift):
    """Escape / unescape writers: 4096-byte blocks without any escape byte take a path of their own (SWAR map, one
    16-byte store); `shift` escape bytes in front move every later block's output to each alignment, '<' (-> FF)
    and FF / 5C (-> 5C FF / 5C 5C) sit at block edges, and a block in the middle is all escapes."""
    rng = random.Random(900 + shift)
    plain = bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz <>,0123456789") for _ in range(5 * 4096))
    data = bytearray(b"\xff" * shift + plain)
    for p in (4095, 4096, 8191, 12288, 12289):
        data[p + shift] = rng.choice(b"\\\xff<")
    data += b"\\\xff" * 2048 + plain[:4096 + 77]
    data = bytes(data)
    c = lz.CompressAsync(data)
    assert c == oracle.lzss_compress(data, 4096)
    assert lz.Decompress(c, False) == data


def test_unescape_parity_across_many_blocks(lz, oracle):
    """Runs of backslashes far longer than an unescape block (4 KiB of escaped stream) and than one lane's chunk of
    block summaries (16 blocks): whether a 5C escapes or is escaped depends on the parity of the run before it,
    carried across blocks by k_une_carry's scan.  Odd and even run lengths, a run that ends the stream."""
    for a, b in ((300001, 70000), (65536 * 3, 65537), (12345, 1)):
        data = b"\\" * a + b"x\xff<" + b"\\" * b + text(a & 255, 50000) + b"\\" * (b // 3)
        c = lz.CompressAsync(data)
        assert lz.Decompress(c) == oracle.lzss_decompress(c) == data


def test_sample_sends_incompressible_streams_to_the_bucket_search(lz, oracle):
    """Large inputs (8192 tiles = 64 MiB and more; here the threshold is lowered to 1024) walk a sample of 64 tiles
    first: all noise -> no chain walk at all; noise with text in the second half -> the sample is split, the chain
    walk runs and hands the noisy strips back.  Below the threshold the same streams take the chain walk unsampled."""
    rng = np.random.default_rng(77)
    n = 9 << 20
    noise = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
    half = noise[: n // 2 + 12345] + text(78, n // 2)
    want = {id(d): oracle.lzss_compress_mt(d, 4096, oracle.host_cores(), 4096) for d in (noise, half)}
    os.environ["RSN_LZSS_SAMPLE_MIN_TILES"] = "1024"
    try:
        for data, walks in ((noise, False), (half, True)):
            c, p = _prof(lz, data)
            if _chain_mode():
                assert p["lzss_sample"][0] == 2                            # the list and the 64-tile walk
                assert ("lzss_match_chain" in p) == walks
            assert c == want[id(data)]
            assert lz.Decompress(c) == data
    finally:
        del os.environ["RSN_LZSS_SAMPLE_MIN_TILES"]
    for data in (noise, half):
        c, p = _prof(lz, data)
        assert "lzss_sample" not in p and c == want[id(data)]


def test_chain_entry_fixed_by_second_look(lz, oracle):
    """A tile whose warm-up chain has not merged with the true chain when it enters the tile: a 200-periodic stretch
    lies across a tile boundary under a 300-byte window (chains of different phase keep their phase inside it), text
    on both sides.  k_chain_verify lists the tile, the second look walks it from the true entry (the tile before's
    exit) and the per-tile chains join: no general parse."""
    rng = random.Random(41)
    period = bytes(rng.randrange(97, 123) for _ in range(200))
    for lead in (8192 - 900, 2 * 8192 - 1500, 8192 - 140):
        data = text(51, lead) + (period * 12)[:2100] + text(52, 30000)
        c, p = _prof(lz, data, 300)
        assert c == oracle.lzss_compress(data, 300)
        assert lz.Decompress(c) == data
        if _chain_mode() and "RSN_LZSS_NO_FUSED_PARSE" not in __import__("os").environ:
            # (with a wavefront per chain -- r03: one chain per wavefront -- every lead needs the second look; with eight chains per
            #  wavefront and a start every 64 positions some of these joints come out right at once)
            assert 1 <= p["lzss_match_chain"][0] <= 4 and "lzss_parse_mark" not in p, sorted(p)


def test_decode_without_unescape_pass_and_its_capacity_contract(lz, oracle):
    """A stream without any 5C byte is unescaped by the emit kernel itself (FF -> '<' on its way into the caller's
    buffer); with a backslash anywhere the separate passes run.  Both honour the size-query contract: a buffer
    that is too small fails with RSN_ERR_CAPACITY and the needed size, exactly E bytes suffice."""
    import torch
    from raisin_amd import _lib
    base = text(61, 200000).replace(b"\\", b"/")                  # holds "<tag>" (-> FF) but no backslash
    for data in (base, base[:100000] + b"\\" + base[100000:]):
        c = oracle.lzss_compress(data)
        src = torch.frombuffer(bytearray(c), dtype=torch.uint8).cuda()
        small = torch.empty(len(data) - 1 - (len(data) - 1) % 16, dtype=torch.uint8, device="cuda")
        with pytest.raises(_lib.RsnError) as ei:
            _lib.call_dev(_lib.lib().rsn_lzss_decompress_dev, src.data_ptr(), src.numel(), small.data_ptr(), small.numel(), None)
        assert ei.value.code == _lib.RSN_ERR_CAPACITY and ei.value.needed >= len(data)
        exact = torch.empty(len(data) + 16 - len(data) % 16, dtype=torch.uint8, device="cuda")[: len(data)]
        got = _lib.call_dev(_lib.lib().rsn_lzss_decompress_dev, src.data_ptr(), src.numel(), exact.data_ptr(), len(data), None)
        assert got == len(data) and bytes(exact.cpu().numpy()) == data


def test_sections_give_the_same_stream(oracle):
    """A stream of 2 GiB and more is encoded section by section (lzss_encode_dev: 32-bit positions): each section's stream begins 64
    tiles before the position the chain enters it -- the exit of the section before -- and ends a window behind its last tile.
    RSN_LZSS_SECTION_MIB=4 forces 4 MiB sections on inputs of 20-30 MiB (in a process of its own: the switch is read once): the bytes
    are those of the single pass and the oracle's (text; sections of noise with escapes, periodic data and runs; a 37-byte period)."""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "import numpy as np, torch\n"
            "import workloads as W\nfrom raisin_amd import lz\n"
            "from oracle import oracle as O\n"
            "g = torch.Generator().manual_seed(3)\n"
            "text = W.config_input('4', 30 << 20)\n"
            "parts = []\n"
            "for k in range(24):\n"
            "    kind = k %% 4\n"
            "    if kind == 0: parts.append(W.config_input('4', 40 << 20)[(k + 3) << 20:(k + 4) << 20])\n"
            "    elif kind == 1: parts.append(torch.randint(0, 256, (1 << 20,), dtype=torch.uint8, generator=g))\n"
            "    elif kind == 2: parts.append(W.config_input('3', 1 << 20))\n"
            "    else: parts.append(torch.randint(97, 101, ((1 << 20) // 37 + 1,), dtype=torch.uint8, generator=g).repeat_interleave(37)[:1 << 20])\n"
            "mixed = torch.cat(parts)\n"
            "per = torch.randint(32, 127, (37,), dtype=torch.uint8, generator=g).repeat((20 << 20) // 37)\n"
            "for name, t in (('text', text), ('mixed', mixed), ('period37', per)):\n"
            "    host = bytes(t.numpy())\n"
            "    out = bytes(lz.compress_tensor(t.cuda()).cpu().numpy())\n"
            "    ok, bad = O.lzss_check(host, out)\n"
            "    print(name, hashlib.sha256(out).hexdigest(), len(out), ok, bad)\n" % root)
    res = []
    for env in ({}, {"RSN_LZSS_SECTION_MIB": "4"}, {"RSN_LZSS_SECTION_MIB": "7", "RSN_LZSS_NO_FUSED_PARSE": "1"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l.split() for l in r.stdout.strip().splitlines()]
        assert len(lines) == 3 and all(l[3] == "True" for l in lines), (env, lines)
        res.append([l[:3] for l in lines])
    assert res[0] == res[1] == res[2]


def test_decode_sections_give_the_same_bytes(oracle):
    """A stream of 4 GiB and more -- compressed or decoded -- is expanded section by section (lzss_decode_sections): cuts at 4 KiB
    block starts of the compressed stream, moved past a token that straddles one; a section is decoded as a stream of its own that
    begins with the escaped bytes before it (the largest back-pointer's worth), and lands on the place those came from.
    RSN_LZSS_DEC_SECTION_MIB=1 / 3 forces sections of 1 and 3 MiB on streams of 8-24 MiB (a process of its own: the switch is read
    once): text; noise with escapes, periodic data, runs and a 37-byte period mixed; a window of 16 (short tokens: many cuts fall
    inside one); streams with 5C / FF / '<' bytes, whose escape pairs the cuts split.  Every result is the input."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "import numpy as np, torch\n"
            "import workloads as W\nfrom raisin_amd import lz\n"
            "g = torch.Generator().manual_seed(5)\n"
            "text = W.config_input('4', 24 << 20)\n"
            "parts = []\n"
            "for k in range(16):\n"
            "    kind = k %% 4\n"
            "    if kind == 0: parts.append(W.config_input('4', 24 << 20)[(k + 3) << 20:(k + 4) << 20])\n"
            "    elif kind == 1: parts.append(torch.randint(0, 256, (1 << 20,), dtype=torch.uint8, generator=g))\n"
            "    elif kind == 2: parts.append(W.config_input('3', 1 << 20))\n"
            "    else: parts.append(torch.randint(97, 101, ((1 << 20) // 37 + 1,), dtype=torch.uint8, generator=g).repeat_interleave(37)[:1 << 20])\n"
            "mixed = torch.cat(parts)\n"
            "esc = torch.tensor([0x5C, 0x3C, 0xFF, 0x5C, 0x5C, 0x41, 0x3C, 0x3C], dtype=torch.uint8)[torch.randint(0, 8, (8 << 20,), generator=g)]\n"
            "for name, t, w in (('text', text, 4096), ('mixed', mixed, 4096), ('text-w16', text[:8 << 20], 16), ('escapes', esc, 4096), ('escapes-w100', esc, 100)):\n"
            "    d = t.cuda()\n"
            "    c = lz.compress_tensor(d, window=w)\n"
            "    back = lz.decompress_tensor(c)\n"
            "    print(name, c.numel(), back.numel() == d.numel() and bool(torch.equal(back, d)), hashlib.sha256(bytes(back.cpu().numpy())).hexdigest())\n" % root)
    res = []
    for env in ({}, {"RSN_LZSS_DEC_SECTION_MIB": "1"}, {"RSN_LZSS_DEC_SECTION_MIB": "3", "RSN_DEBUG": "1"}, {"RSN_LZSS_DEC_SECTION_MIB": "2", "RSN_LZSS_DEC_JUMP": "1"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l.split() for l in r.stdout.strip().splitlines()]
        assert len(lines) == 5 and all(l[2] == "True" for l in lines), (env, lines)
        if env.get("RSN_DEBUG"): assert r.stderr.count("lzss decode section") >= 10, r.stderr[-2000:]
        res.append(lines)
    assert res[0] == res[1] == res[2] == res[3]


def test_a_storm_of_large_calls_queues_instead_of_failing(oracle):
    """every calling thread has a scratch arena of its own and a 1 GiB LZSS encode wants ~12 GiB of it; concurrent large
    calls are admitted while their needs fit the device (RSN_SCRATCH_GIB: here 1.5 GiB, so that 64 MiB calls -- ~1 GiB each -- run one
    at a time) and the rest wait; a call that finishes while others wait hands its buffers back.  Six threads, all results the
    single-threaded bytes; the log shows that calls did wait."""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, hashlib, threading; sys.path.insert(0, %r)\n"
            "import workloads as W\nfrom raisin_amd import lz\n"
            "datas = [bytes(W.config_input('4', (64 << 20) + k * 4099).numpy()) for k in range(3)]\n"
            "want = [hashlib.sha256(lz.CompressAsync(d)).hexdigest() for d in datas]\n"
            "got = [None] * 6\n"
            "back = [None] * 6\n"
            "def run(i):\n"
            "    c = lz.CompressAsync(datas[i %% 3])\n"
            "    got[i] = hashlib.sha256(c).hexdigest()\n"
            "    back[i] = lz.Decompress(c) == datas[i %% 3]\n"                # (the decoder's scratch goes through the same gate)
            "ts = [threading.Thread(target=run, args=(i,)) for i in range(6)]\n"
            "[t.start() for t in ts]; [t.join() for t in ts]\n"
            "print('OK' if got == want + want and all(back) else 'MISMATCH', got, want, back)\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, RSN_SCRATCH_GIB="1.5", RSN_SCRATCH_DEBUG="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.strip().startswith("OK"), r.stdout[-2000:]
    assert "waits" in r.stderr, r.stderr[-2000:]
    # (the admission covers the host-buffer call's STAGING too -- n + 1.125 x bound next to the encoder's 15 n: a 64 MiB call
    #  states 1.14 GiB, not the 0.94 of the codec alone)
    import re
    needs = [float(x) for x in re.findall(r"needs ([0-9.]+) GiB of scratch", r.stderr)]
    assert needs and max(needs) >= 1.10, needs


def _periodic_cases():
    rng = np.random.default_rng(606)
    blk = bytes(rng.integers(0, 254, size=4096, dtype=np.uint8).tolist())
    blk = blk.replace(b"\\", b"a").replace(b"\xff", b"b")               # (config 3's block: nothing that needs an escape; '<' may occur)
    blk = blk[:100] + b"<" + blk[101:2000] + b"<<" + blk[2002:]
    assert len(blk) == 4096
    small = bytes(rng.integers(97, 123, size=256, dtype=np.uint8).tolist())
    return {
        "config 3's shape": (blk * 700, 4096),
        "a remainder that goes out as a token": (blk * 300 + blk[:1234], 4096),
        "a remainder of three raw bytes, one of them a '<'": (blk * 300 + blk[:99] + b"<ab", 4096),
        "a remainder of one byte": (blk * 260 + blk[:1], 4096),
        "text in front of the period": (text(9, 200000) + blk * 400, 4096),
        "a head that ends inside a tile": (text(10, 70001) + blk * 400 + blk[:77], 4096),
        "window 256, period 256": (small * 9000, 256),
        "window 1024, period 256 (the period divides the window)": (small * 9000, 1024),
        "the period breaks in the last quarter (no tail there)": (blk * 300 + b"!" + blk * 100, 4096),
        "two periods, the second to the end": (blk * 100 + blk[::-1] * 300, 4096),
        "period 4096 under window 2048 (nothing repeats inside the window)": (blk * 60, 2048),
    }


def test_periodic_tail_is_the_oracles_stream(lz, oracle):
    """r06: a stream that repeats with the window's length from some chunk on is encoded as its head + arithmetic (k_periodic_tail)
    and decoded as its head + out[q] = out[q - P] (k_lzd_run_fill).  Both against the oracle, byte for byte, over the shapes the
    arithmetic has to get right: remainders that become a token / raw bytes / nothing, '<' in the repeated block, heads of text,
    other windows, a break late in the stream, a second period."""
    for name, (data, w) in _periodic_cases().items():
        want = oracle.lzss_compress(data, w)
        got = lz.CompressAsync(data, False, w)
        assert got == want, name
        assert lz.Decompress(got) == data, name


def test_periodic_tail_switches_give_the_same_bytes(oracle):
    """... and the same streams through the whole pipeline / the ordinary decoder (RSN_LZSS_NO_PERIODIC_TAIL, RSN_LZSS_DEC_NO_RUN_TAIL:
    read once per process), plus hand-written streams that end in a token run only a foreign encoder writes: periods that are not
    the window, a run in front of which nothing stands (an error either way), a last item that is a token with another distance."""
    import hashlib
    import pickle
    import subprocess
    import sys
    import tempfile
    cases = _periodic_cases()
    rng = np.random.default_rng(33)
    lit = bytes(rng.integers(97, 123, size=70000, dtype=np.uint8).tolist())
    foreign = [lit + b"<5,5>" * 300000, lit + b"<4096,4096>" * 400 + b"<777,700>", lit + b"<8192,8192>" * 300 + b"tail\xff!",
               lit[:3000] + b"<3000,3000>" * 500, lit + b"<7,7>" * 10 + lit[:50] + b"<100,100>" * 20000 + b"<100,3>",
               lit + b"<9000,9000>" * 200]                                 # (a period above what the fill kernel keeps: the ordinary path)
    bad = [b"<4096,4096>" * 100000, lit[:100] + b"<4096,4096>" * 100000]  # the run's first token points before the data (lzss.go:350)
    code = ("import sys, hashlib, pickle; sys.path.insert(0, %r)\nfrom raisin_amd import lz, RsnError\n"
            "datas, streams, bad = pickle.load(open(sys.argv[1], 'rb'))\n"
            "out = [hashlib.sha256(lz.CompressAsync(d, False, w)).hexdigest() for d, w in datas] + [hashlib.sha256(lz.Decompress(c)).hexdigest() for c in streams]\n"
            "for b in bad:\n"
            "    try:\n        lz.Decompress(b); out.append('no error')\n"
            "    except RsnError as e:\n        out.append('error %%d' %% e.code)\n"
            "print('|'.join(out))\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    streams = [oracle.lzss_compress(d, w) for d, w in cases.values()] + foreign
    want = [hashlib.sha256(c).hexdigest() for c in streams[:len(cases)]] + [hashlib.sha256(oracle.lzss_decompress(c)).hexdigest() for c in streams]
    want += ["error -3"] * len(bad)
    with tempfile.NamedTemporaryFile(suffix=".pkl") as f:
        pickle.dump((list(cases.values()), streams, bad), f)
        f.flush()
        for env in ({}, {"RSN_LZSS_NO_PERIODIC_TAIL": "1", "RSN_LZSS_DEC_NO_RUN_TAIL": "1"}):
            out = subprocess.run([sys.executable, "-c", code, f.name], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
            assert out.returncode == 0, out.stderr[-2000:]
            assert out.stdout.strip().splitlines()[-1].split("|") == want, (env, out.stdout[-600:])


def test_threads_with_different_tail_lengths_and_windows(lz, oracle):
    """The kernels that take dynamic LDS (k_lzd_compose: four bytes per position of the longest back-pointer's tail; k_match2: by window)
    have that maximum set per FUNCTION, for all threads: one thread decoding streams with 16 000-byte pointers beside one with 100-byte
    pointers, one encoding under window 8192 beside one under 5000 -- each must keep getting its own answer (r06: the attribute was
    tracked per thread and could be lowered under another thread's next launch)."""
    import threading
    rng = np.random.default_rng(8)
    lit = rng.integers(97, 123, size=40000, dtype=np.uint8).tobytes()
    big = lit + b"<16000,16000>" * 40 + b"<15000,900>" * 300
    small = lit[:3000] + b"<100,100>" * 20000 + b"<37,30>" * 9000
    want_big, want_small = oracle.lzss_decompress(big), oracle.lzss_decompress(small)
    data = text(77, 90000) + b"ab" * 3000 + text(78, 20000)
    want_w = {w: oracle.lzss_compress(data, w) for w in (8192, 5000)}
    errors = []

    def dec(stream, want, tag):
        for r in range(12):
            if lz.Decompress(stream) != want:
                errors.append((tag, r))

    def enc(w):
        for r in range(3):
            if lz.CompressAsync(data, False, w) != want_w[w]:
                errors.append(("window", w, r))
    ts = [threading.Thread(target=dec, args=(big, want_big, "big")), threading.Thread(target=dec, args=(small, want_small, "small")),
          threading.Thread(target=enc, args=(8192,)), threading.Thread(target=enc, args=(5000,))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def _sparse(seed, n, every, hi=128):
    rng = np.random.default_rng(seed)
    a = np.zeros(n, dtype=np.uint8)
    idx = rng.integers(0, n, size=max(1, n // every))
    a[idx] = rng.integers(1, hi, size=idx.size, dtype=np.uint8)
    return a.tobytes()


def _byte_runs(seed, n, longest, alphabet=b"abcdefgh<\\ \n"):
    rng = random.Random(seed)
    out, size = [], 0
    while size < n:
        k = rng.randint(1, longest)
        out.append(bytes([rng.choice(alphabet)]) * k)
        size += k
    return b"".join(out)[:n]


@pytest.mark.parametrize("name,data", [
    ("a byte in 100 set", _sparse(1, 1 << 20, 100)),
    ("a byte in 300 set", _sparse(2, 600000, 300)),
    ("a byte in 3000 set", _sparse(3, 1 << 20, 3000)),
    ("a byte in 3000 set, of two values", _sparse(4, 1 << 20, 3000, hi=3)),          # runs that end alike: the candidate whose run ends with the position's goes on behind it
    ("a byte in 700 set, of two values", _sparse(5, 700001, 700, hi=3)),
    ("runs up to 1000", _byte_runs(6, 1 << 20, 1000)),
    ("runs up to 5000", _byte_runs(7, 1 << 20, 5000)),
    ("runs up to 600 of two bytes", _byte_runs(8, 500000, 600, b"ab")),
    ("runs up to 3000 of two bytes", _byte_runs(9, 800000, 3000, b"a<")),
    ("text, zeros, text", text(10, 100000) + bytes(300000) + text(11, 50000) + bytes(5000) + text(12, 30000)),
    ("zeros with a word now and then", b"".join(bytes(n) + w for n, w in zip(np.random.default_rng(13).integers(1, 2500, size=600).tolist(), [b"word", b"zero", b"\0x\0", b"longer words here"] * 150))),
    ("a run that ends the stream", text(14, 20000) + b"z" * 9000),
    ("a run that begins it", b"z" * 9000 + text(15, 20000)),
], ids=lambda v: v if isinstance(v, str) else "")
def test_byte_runs_are_resolved_in_the_walk(lz, oracle, name, data):
    """r06: a position in a run of one byte with HLMAX of it ahead has thousands of candidates that agree further than the stage reaches;
    the walk resolves them from the window's runs instead of handing the strip to the sweep.  The oracle's bytes."""
    c = lz.CompressAsync(data)
    assert c == oracle.lzss_compress(data)
    assert lz.Decompress(c) == data


def _sorted_lines(seed, n, n_words, longest):
    rng = random.Random(seed)
    words = ["".join(rng.choice("abcdefghij") for _ in range(rng.randint(1, longest))) for _ in range(n_words)]
    lines = sorted(rng.choice(words) + rng.choice(words) for _ in range(n // 4))
    return "\n".join(lines).encode()[:n]


@pytest.mark.parametrize("name,data", [
    ("sorted lines, many alike", _sorted_lines(1, 1 << 20, 300, 3)),
    ("sorted lines, fewer alike", _sorted_lines(2, 1 << 20, 3000, 9)),
    ("sorted lines of one and two letters", _sorted_lines(3, 700000, 40, 2)),
    ("a line repeated, then one a byte longer", (b"ab\n" * 400 + b"aba\n" * 300 + b"abab\n" * 250 + b"abb\n" * 500) * 40),
    ("period 256 broken now and then", bytes(bytearray((bytes(range(32, 127)) * 3)[:256] * 4096)[:1 << 20])),
], ids=lambda v: v if isinstance(v, str) else "")
def test_contenders_of_the_farthest_long_candidate_are_followed(lz, oracle, name, data):
    """r06: more long candidates than a visit lists, and the farthest does not run to its limit: the candidates that can outlast it
    (its byte at the mismatch, the first eight bytes, the stage's last eight) are followed to their ends, up to eight of them."""
    if name.startswith("period 256"):
        b = bytearray(data)
        rng = random.Random(4)
        at = 30000
        while at < len(b):
            b[at] = ord(rng.choice("ABCDEFG")); at += rng.randint(20000, 200000)
        data = bytes(b)
    c = lz.CompressAsync(data)
    assert c == oracle.lzss_compress(data)
    assert lz.Decompress(c) == data


def _lines_mostly_alike(seed, n, common, share, alphabet="0123456789,.x"):
    rng = random.Random(seed)
    out, size = [], 0
    while size < n:
        if rng.random() < share:
            ln = common
        else:
            ln = "".join(rng.choice(alphabet) for _ in range(rng.randint(1, 2 * len(common)))) + "\n"
        out.append(ln); size += len(ln)
    return "".join(out).encode()[:n]


def _line_repeated(seed, n, lines, longest):
    rng = random.Random(seed)
    out, size = [], 0
    while size < n:
        k = rng.randint(1, longest)
        ln = rng.choice(lines)
        out.append(ln * k); size += len(ln) * k
    return "".join(out).encode()[:n]


_LOG = ["worker idle\n", "heartbeat ok 200\n", "retrying connection to 10.0.0.7\n", "a" * 63 + "\n", "x\n", "ab\n",
        "GET /index.html HTTP/1.1 200 5120 \"-\" \"Mozilla/5.0 (X11; Linux x86_64) AppleWebKit/537.36\"\n"]


@pytest.mark.parametrize("name,data", [
    ("csv, nine rows in ten the same", _lines_mostly_alike(1, 1 << 20, "0,0,0,0.0,\n", 0.9)),
    ("csv, 99 rows in 100 the same", _lines_mostly_alike(2, 1 << 20, "0,0,0,0.0,\n", 0.99)),
    ("csv, half the rows the same", _lines_mostly_alike(3, 700000, "1,2,3\n", 0.5)),
    ("a two-byte line, nine in ten", _lines_mostly_alike(4, 600000, "0\n", 0.9, "01")),
    ("a 64-byte row, 19 in 20", _lines_mostly_alike(5, 1 << 20, "0123456789abcdef" * 3 + "0123456789abcde\n", 0.95)),
    ("log lines repeated up to 300 times", _line_repeated(6, 1 << 20, _LOG, 300)),
    ("log lines repeated up to 20 times", _line_repeated(7, 1 << 20, _LOG, 20)),
    ("log lines repeated up to 3000 times", _line_repeated(8, 1 << 20, _LOG[:4], 3000)),
    ("the same line with < and a backslash in it", _line_repeated(9, 500000, ["<a href=\\x>\n", "<b>\n", "\\\\\n"], 200)),
    ("short period, then text, then the period again", (b"abcabd" * 3000 + text(20, 30000) + b"abcabd" * 2000 + text(21, 5000)) * 6),
    ("periods 2, 3, 5, 7, 11, 13 in turn", b"".join((bytes(range(97, 97 + p)) * (9000 // p + i)) for i in range(40) for p in (2, 3, 5, 7, 11, 13))),
    ("a period of 64 and one of 65", (bytes(range(48, 112)) * 300 + bytes(range(48, 113)) * 300) * 10),
    ("utf-16 zeros between letters, lines repeated", _line_repeated(10, 600000, ["ok\n", "fail\n"], 500).decode().encode("utf-16-le")),
], ids=lambda v: v if isinstance(v, str) else "")
def test_repeated_lines_are_resolved_in_the_walk(lz, oracle, name, data):
    """r06: a position in a stretch of a short period -- a line or a record repeated -- is resolved from the window's stretches
    (chain_period_visit); the lean instance of the walk gives such tiles up and the walk is done again by the other.
    The oracle's bytes."""
    c = lz.CompressAsync(data)
    assert c == oracle.lzss_compress(data)
    assert lz.Decompress(c) == data
