"""CPU tests: the C-ABI library loads and exports every symbol include/rsn.h declares,
fails loudly without a device, and its HOST logic (Go-exact tree, header) matches the oracle."""
import ctypes
import os
import random
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from raisin_amd import _lib
    return _lib


def test_every_declared_symbol_is_exported(built):
    hdr = open(os.path.join(ROOT, "include", "rsn.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rsn_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    L = built.lib()
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert declared == set(built.SYMBOLS)


def test_nothing_but_the_declared_symbols_is_exported(built):
    """the dynamic symbol table is rsn.h's set and nothing else (built with -fvisibility=hidden;
    a host binary's own `env_int` must not interpose the library's).  The HIP runtime's registration objects
    (__hip_*) are the one allowed exception."""
    import subprocess
    hdr = open(os.path.join(ROOT, "include", "rsn.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rsn_[a-z0-9_]+)\s*\(", hdr))
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "raisin_amd", "librsn.so")], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    extra = {x for x in exported - declared if not x.startswith("__hip_")}
    assert not extra, sorted(extra)[:20]
    assert declared <= exported


def test_no_cpu_fallback(built):
    """Without a HIP device every codec entry point must fail loudly (never compute on the CPU)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from raisin_amd import RsnError, huffman, lz
    for fn in (lambda: huffman.Compress(b"abc"), lambda: huffman.Decompress(b"1|a\\\n\x00"),
               lambda: lz.CompressAsync(b"abc"), lambda: lz.Decompress(b"abc")):
        with pytest.raises(RsnError) as e:
            fn()
        assert e.value.code == -4 and "no CPU fallback" in str(e.value)


def test_batch_entry_point_checks_its_arguments_before_any_device(built):
    """rsn_huffman_compress_batch: null arrays and an empty chunk are refused before a device is looked for (an empty
    chunk as the single call refuses it, huffman.go:102); with good arguments and no device it fails like the single call,
    and every outs[i] is NULL either way."""
    import ctypes
    import torch
    from raisin_amd import _lib
    L = _lib.lib()
    k = 3
    bufs = [b"abc" * 10, b"", b"xyz" * 5]
    ins = (ctypes.c_char_p * k)(*bufs)
    lens = (ctypes.c_size_t * k)(*[len(b) for b in bufs])
    outs = (ctypes.POINTER(ctypes.c_uint8) * k)()
    olens = (ctypes.c_size_t * k)()
    assert L.rsn_huffman_compress_batch(k, None, lens, outs, olens) == -1
    assert L.rsn_huffman_compress_batch(k, ins, lens, outs, olens) == -2 and b"empty" in L.rsn_last_error()
    assert all(not outs[i] for i in range(k)) and all(olens[i] == 0 for i in range(k))
    if not torch.cuda.is_available():
        lens[1] = 1
        ins[1] = b"q"
        assert L.rsn_huffman_compress_batch(k, ins, lens, outs, olens) == -4 and b"no CPU fallback" in L.rsn_last_error()
        assert all(not outs[i] for i in range(k))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "raisin_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in txt.replace("rsn_oracle", "oracle") or f == "__init__.py" and False, (dirpath, f)


def _counts(data, oracle):
    runes = oracle.utf8_runes(data)
    vals, cnts = np.unique(runes, return_counts=True)
    return {int(v): int(c) for v, c in zip(vals, cnts)}


def test_host_tree_and_header_match_oracle(built, oracle, samiam):
    from raisin_amd import huffman
    rng = random.Random(12)
    cases = [samiam, b"Hello world!\n", b"ab", b"a\nb\\", b"AB\\\\A", "héllo ✓ 𝄞\n".encode() * 7,
             bytes(rng.randrange(256) for _ in range(20000)),
             bytes(rng.choice(b"aaaaabbbc") for _ in range(5000)),
             bytes(rng.randrange(128) for _ in range(100000))]
    for data in cases:
        table, header = huffman.plan(_counts(data, oracle))
        want = oracle.huffman_table(data)
        assert table == [(r, c, l) for r, f, c, l in want]
        ref = oracle.huffman_compress(data)
        assert header == ref[:ref.index(b"\\\n")]
        assert huffman.parse_header(header) == sorted(_counts(data, oracle).items())


def test_heap_tie_breaks_on_equal_frequencies(built, oracle):
    """All-equal counts exercise nothing but Go's container/heap sift order."""
    from raisin_amd import huffman
    for k in (2, 3, 5, 6, 7, 12, 33, 100, 128):
        data = bytes(range(k)) * 3
        table, _ = huffman.plan({i: 3 for i in range(k)})
        assert table == [(r, c, l) for r, f, c, l in oracle.huffman_table(data)]


def test_large_alphabet_tree_matches_oracle(built, oracle):
    """>= 4096 symbols takes the radix leaf sort and the packed heap; counts scaled past 2**32
    take the wide heap and must give the same tree (every comparison scales with them)."""
    import random
    from raisin_amd import huffman
    rng = random.Random(11)
    runes = rng.sample([r for r in range(0x20, 0x30000) if not 0xD800 <= r < 0xE000 and r != 0x5C], 6000)
    counts = {r: rng.choice((1, 1, 2, 3, 7)) for r in runes}
    text = [r for r, c in counts.items() for _ in range(c)]
    rng.shuffle(text)
    data = "".join(map(chr, text)).encode("utf-8")
    want = [(r, c, l) for r, f, c, l in oracle.huffman_table(data)]
    table, header = huffman.plan(counts)
    assert table == want
    assert huffman.parse_header(header) == sorted(counts.items())
    wide, _ = huffman.plan({r: c << 33 for r, c in counts.items()})
    assert wide == want
    # (r06: the sifts take three levels a step, frequencies and ids in arrays of their own) a table of 2b's shape: most runes once or
    # twice, a few thousand around sixty times, a few often -- sixteen levels of heap
    runes = rng.sample([r for r in range(0x20, 0x10FFFF) if not 0xD800 <= r < 0xE000 and r != 0x5C], 60000)
    counts = {r: (1 if i < 42000 else 2 if i < 48000 else 3 if i < 49000 else rng.randint(40, 90) if i < 59800 else rng.randint(5000, 9000)) for i, r in enumerate(runes)}
    text = [r for r, c in counts.items() for _ in range(c)]
    rng.shuffle(text)
    data = "".join(map(chr, text)).encode("utf-8")
    want = [(r, c, l) for r, f, c, l in oracle.huffman_table(data)]
    assert huffman.plan(counts)[0] == want
    assert huffman.plan({r: c << 33 for r, c in counts.items()})[0] == want


def test_header_parse_quirks(built):
    from raisin_amd import RsnError, huffman
    assert huffman.parse_header(b"3|\\n5||7|9") == [(10, 3), (0x39, 7), (0x7C, 5)]     # '\\n' escape, '|' and digit symbols
    assert huffman.parse_header("2|é1|x".encode()) == [(ord("x"), 1), (ord("é"), 2)]  # continuation bytes are skipped (huffman.go:222)
    assert huffman.parse_header(b"|a") == [(ord("a"), 0)]                             # Atoi("") == 0 (huffman.go:207)
    assert huffman.parse_header(b"1|a5|a") == [(ord("a"), 5)]                         # later entry overwrites
    for bad in (b"3|", b"3|\\"):                                                       # index out of range (huffman.go:210)
        with pytest.raises(RsnError):
            huffman.parse_header(bad)
    with pytest.raises(RsnError):
        huffman.plan({})                                                               # heap.Pop on an empty heap (huffman.go:102)


def test_engine_mirror_host_side(built):
    from raisin_amd import engine, lz
    assert engine.ByteCountSI(1000) == "1.0 kB" and engine.ByteCountSI(10) == "10 B"   # engine/util_test.go:7-17
    assert engine.ByteCountSI(1234567) == "1.2 MB"
    assert engine.parseAlgorithms("lzss,arithmetic,huffman,[lzss,arithmetic],gzip") == [
        ["lzss"], ["arithmetic"], ["huffman"], ["lzss", "arithmetic"], ["gzip"]]       # cmd/cli.go:169,203-231
    assert set(engine.Writers) == set(engine.Readers) == {"lzss", "huffman"}
    with pytest.raises(KeyError):
        engine.CompressedFile(CompressionEngine="dmc").Write(b"x")
    with pytest.raises(ValueError):
        lz.NewWriterLevel(None, -1)                                                     # lzss.go:43-45
    assert lz.DefaultWindowSize == 4096


def test_time_taken_column_is_gos_duration_string(built):
    """engine.go:425: duration.Round(10*time.Microsecond).String() -- README.md:153-167 shows "190µs", ai/data.json "4.61328s";
    the same nanoseconds through the Python mirror and the C++ host."""
    import subprocess
    from raisin_amd import engine
    want = {0: "0s", 4999: "0s", 5000: "10\u00b5s", 190000: "190\u00b5s", 194999: "190\u00b5s", 195000: "200\u00b5s", 1234567: "1.23ms",
            999995000: "1s", 1500000000: "1.5s", 4613280000: "4.61328s", 60000000000: "1m0s", 61230000000: "1m1.23s",
            3600000000000: "1h0m0s", 3723500000000: "1h2m3.5s"}
    exe = os.path.join(ROOT, "raisin_amd", "host", "rsn")
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(exe)])
    for ns, text in want.items():
        assert engine._go_duration_ns(engine._go_round_ns(ns, 10000)) == text, ns
        assert subprocess.check_output([exe, "-fmtduration=%d" % ns]).decode("utf-8").strip() == text, ns
    assert engine._go_duration(60.0) == "1m0s" and engine._time_taken(0.00019) == "190\u00b5s"   # README.md:157


def test_legacy_lz_compress_host_only(built, oracle, samiam, known):
    """(f)#4: lz.Compress (lzss.go:224) lives in librsn as host code of its own -- README's 21-byte answer
    (README.md:165), the ai/data.json sizes, and the oracle's independent restatement; needs no device."""
    from raisin_amd import lz
    ref = known["reference"]
    assert lz.Compress(b"abc" * 8 + b"\n") == b"abcabca<6,6>b<12,10>\n"
    assert len(lz.Compress(b"abc" * 8 + b"\n")) == ref["lzss_legacy_abc_size"]
    assert len(lz.Compress(b"a" * 100000)) == ref["lzss_legacy_aaa_100000_size"]
    assert len(lz.Compress((b"abcdefghijklmnopqrstuvwxyz" * 3847)[:100000])) == ref["lzss_legacy_alphabet_100000_size"]
    assert lz.Compress(b"a") == b"a" and lz.Compress(b"") == b""
    rng = np.random.default_rng(3)
    cases = [samiam, samiam * 3, b"Hello world!\n", bytes(rng.integers(0, 256, 9000, dtype=np.uint8)),
             bytes(rng.integers(60, 64, 9000, dtype=np.uint8)), b"<\\\xff" * 700, b"ab" * 5000]
    for d in cases:
        for w in (4096, 8192, 100, 0):
            assert lz.Compress(d, False, w) == oracle.lzss_compress_legacy(d, w)


def test_legacy_lz_compress_refuses_what_would_run_for_hours(built):
    """lz.Compress is the reference's O(n * window) serial loop on the calling thread (O(n^2) for window <= 0) and a cgo
    call cannot be interrupted: above 64 MiB (1 MiB for unbounded / huge windows) the entry point returns RSN_ERR_LIMIT."""
    from raisin_amd import _lib
    L = _lib.lib()
    out = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_size_t(0)
    big = bytes((1 << 20) + 1)
    assert L.rsn_lzss_compress_legacy(big, len(big), 0, ctypes.byref(out), ctypes.byref(n)) == -6 and b"legacy" in L.rsn_last_error()
    assert L.rsn_lzss_compress_legacy(big, len(big), 1 << 20, ctypes.byref(out), ctypes.byref(n)) == -6
    assert L.rsn_lzss_compress_legacy(big[:1 << 20], 1 << 20, 64, ctypes.byref(out), ctypes.byref(n)) == 0 and n.value > 0
    L.rsn_free(out)


def test_host_code_under_address_and_ub_sanitizers(tmp_path):
    """The host-side code of librsn (Go-exact heap, header writer / parser, code assignment, the legacy lz.Compress) built
    with -fsanitize=address,undefined and driven over a few hundred alphabets (up to 50 000 symbols, 64-bit counts,
    '\\\\' moved to the front) and inputs.  Sanitizers exist for the CPU build only; the kernels are covered by parity."""
    import subprocess
    src = os.path.join(ROOT, "raisin_amd", "csrc")
    exe = str(tmp_path / "host_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I" + src,
           os.path.join(ROOT, "tests", "host_sanitize_test.cpp"), os.path.join(src, "huff_host.cpp"), os.path.join(src, "lzss_legacy.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "host sanitizer run ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def _reference_checkout(tmp_path):
    import shutil
    ref = "/root/reference/compressor"
    if not os.path.isdir(ref):
        pytest.skip("the reference is not on this machine (it never ships to the GPU box)")
    root = tmp_path / "raisin"
    for pkg in ("lz", "huffman"):
        shutil.copytree(os.path.join(ref, pkg), root / "compressor" / pkg)
    for dirpath, _, files in os.walk(root):
        os.chmod(dirpath, 0o755)
        for f in files:
            os.chmod(os.path.join(dirpath, f), 0o644)
    return root


def _split_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("rsn_split", os.path.join(ROOT, "go", "overlay", "split.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_cgo_recipe_splits_the_reference_packages(tmp_path):
    """INTEGRATION.md step 1 as a program (VERDICT r3 task 1a): go/overlay/split.py on a copy of the reference's two packages.
    Under both tags every package-level identifier used is defined once, no import is orphaned (lzss.go's `sync` goes with
    CompressAsync), compressorWorker stays where CompressRecursive (lzss.go:203) sees it, and no line is lost."""
    root = _reference_checkout(tmp_path)
    S = _split_module()
    assert S.main([str(root), "--dry-run"]) == 0
    assert S.main([str(root)]) == 0
    lz = (root / "compressor" / "lz" / "lzss.go").read_text()
    pure = (root / "compressor" / "lz" / "lzss_purego.go").read_text()
    assert "func compressorWorker(" in lz and "func CompressRecursive(" in lz and '"sync"' not in lz
    assert pure.startswith("//go:build !rsn\n// +build !rsn\n") and '"sync"' in pure
    for f in ("func CompressAsync(", "func compressorWorkerAsync(", "func Compress(", "func Decompress("):
        assert f in pure and f not in lz
    assert (root / "compressor" / "lz" / "lzss_rsn.go").read_text().startswith("//go:build rsn\n// +build rsn\n")
    hp = (root / "compressor" / "huffman" / "huffman_purego.go").read_text()
    hf = (root / "compressor" / "huffman" / "huffman.go").read_text()
    assert "func Compress(" in hp and "func Decompress(" in hp and "func Compress(" not in hf and "func NewWriter(" in hf
    with pytest.raises(SystemExit):              # a second run finds nothing to move and says so
        S.main([str(root)])


def test_cgo_recipe_check_catches_the_r3_defects(tmp_path):
    """The check is not vacuous: the recipe VERDICT r3 faulted (compressorWorker moved with CompressAsync) is reported, and so is
    an import left behind."""
    root = _reference_checkout(tmp_path)
    S = _split_module()
    bad = dict(S.PLAN["lz"], move=["CompressAsync", "compressorWorkerAsync", "compressorWorker", "Compress", "Decompress"])
    problems = S.process(str(root), "lz", bad, os.path.join(ROOT, "go", "overlay"), True)
    assert any("-tags rsn" in p and "compressorWorker" in p for p in problems), problems
    src = (root / "compressor" / "lz" / "lzss.go").read_text()
    kept, moved = S.split_source(src, S.PLAN["lz"]["move"], "lz")
    kept_with_sync = kept.replace('\t"strconv"\n', '\t"strconv"\n\t"sync"\n', 1)
    _, _, imps = S.import_block(src)
    probs = S.check_build({"lzss.go": kept_with_sync, "lzss_purego.go": moved}, set(S.package_level_defs(src)), {i for i, _ in imps}, "lz")
    assert any("imports sync and does not use it" in p for p in probs), probs


def test_slice_cuts_fall_on_rune_starts(built):
    """rsn_huffman_slice_cuts (host logic of rsn_huffman_compress_sharded, no device needed): every cut is a rune start of Go's decoding
    of the whole input (the third, pure-Python statement of it in test_oracle.go_runes gives the starts), so no UTF-8 sequence is split
    and the slices' runes concatenate to the input's; cuts are strictly increasing, end at n, and short inputs get fewer slices."""
    from test_oracle import go_runes
    L = built.lib()
    rng = random.Random(5)
    alphabet = [b"a", b"z", "é".encode(), "€".encode(), "𝄞".encode(), b"\x80", b"\xE2\x82", b"\xF0\x9D", b"\xFF", b"\xC3", b"\xBF" * 4, b"\xED\xA0\x80"]
    for it in range(300):
        data = b"".join(rng.choice(alphabet) for _ in range(rng.randrange(1, 1500)))
        G = rng.randrange(1, 50)
        cuts = (ctypes.c_size_t * (G + 1))()
        S = L.rsn_huffman_slice_cuts(data, len(data), G, cuts, G + 1)
        assert 1 <= S <= G and S <= max(1, len(data) // 64)
        cs = [cuts[i] for i in range(S + 1)]
        assert cs[0] == 0 and cs[-1] == len(data) and all(a < b for a, b in zip(cs, cs[1:]))
        runes = go_runes(data)
        for a, b in zip(cs, cs[1:]):                      # the slices decode, each on its own, to the input's runes in order
            k = len(go_runes(data[a:b]))
            assert go_runes(data[a:b]) == runes[:k], (it, a, b)
            runes = runes[k:]
        assert runes == []
    assert L.rsn_huffman_slice_cuts(b"", 0, 3, (ctypes.c_size_t * 4)(), 4) == -2
