"""The pipelined host-buffer Huffman decode (rsn_api.hip: huffman_decompress_piped): upload, slice-by-slice decode and
download overlapped -- the same bytes as the serial call (RSN_HOST_SERIAL=1), which are the oracle's."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs():
    import workloads as W
    n = 40 << 20                                   # above the pipeline's 32 MiB threshold: two slices of 32 MiB of payload and a rest
    rng = np.random.default_rng(5)
    return {
        "flat": bytes(W.config_input("2a", n).numpy()),                       # k_dec_flat in slices
        "skewed": bytes(W.config_input("skewed", n + 12345).numpy()),         # the general kernels: a slice starts where the last codeword ended
        "text": bytes(W.config_input("4", n).numpy()),
        "runes": ("Жук €\U0001F600 " * ((n // 3) // 17)).encode(),   # 2-, 3- and 4-byte symbols: output bytes != symbols
        "two": bytes(rng.integers(0, 2, size=n, dtype=np.uint8) + 65),        # one-bit codes: 8 symbols a payload byte, slices of 256 MiB of output... (n symbols)
    }


def test_pipelined_decode_is_the_serial_decode(oracle):
    from raisin_amd import huffman
    datas = _inputs()
    comp = {k: huffman.Compress(v) for k, v in datas.items()}
    for k, v in datas.items():
        assert huffman.Decompress(comp[k]) == v, k                            # pipelined (the default above 32 MiB)
    # a header that announces HALF of what the payload holds (every count even: the same tree): the pipeline's buffer is too small,
    # the serial call decodes it -- and one that announces double: decodes in the pipeline, to the payload's symbols
    d2 = datas["skewed"][: 17 << 20] * 2
    c2 = huffman.Compress(d2)
    sep = c2.index(b"\\\n")
    ents, _ = oracle.header_entries(c2)
    assert all(int(f) % 2 == 0 for f, _ in ents)

    ents = sorted(ents, key=lambda e: e[1] != b"\\")                          # (the backslash's entry first: last, it would run into the separator, huffman.go:210)

    def rewrite(scale):
        hdr = b"".join(str(int(int(f) * scale)).encode() + b"|" + (b"\\n" if sym == b"\n" else sym) for f, sym in ents)
        return hdr + c2[sep:]
    assert huffman.Decompress(rewrite(0.5)) == d2
    assert huffman.Decompress(rewrite(2)) == d2
    # the serial call in a process of its own: the same bytes
    code = ("import sys, hashlib, pickle; sys.path.insert(0, %r)\nfrom raisin_amd import huffman\n"
            "comp = pickle.load(open(sys.argv[1], 'rb'))\n"
            "print(' '.join(k + ':' + hashlib.sha256(huffman.Decompress(c)).hexdigest() for k, c in sorted(comp.items())))\n" % ROOT)
    import pickle
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".pkl") as f:
        pickle.dump(comp, f)
        f.flush()
        out = subprocess.run([sys.executable, "-c", code, f.name], capture_output=True, text=True, timeout=600, env=dict(os.environ, RSN_HOST_SERIAL="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    want = " ".join(k + ":" + hashlib.sha256(v).hexdigest() for k, v in sorted(datas.items()))
    assert out.stdout.strip().splitlines()[-1] == want


def test_pipelined_decode_reports_a_truncated_payload(oracle):
    """A payload cut inside a codeword: RSN_ERR_FORMAT from the last slice, like the serial call; a cut header: the serial call's message."""
    from raisin_amd import RsnError, huffman
    import workloads as W
    data = bytes(W.config_input("skewed", 36 << 20).numpy())
    c = huffman.Compress(data)
    for cut in (len(c) - 1, len(c) - (3 << 20) - 1):
        bad = c[:cut]
        try:
            got = huffman.Decompress(bad)                                     # (a cut that happens to fall on a codeword boundary decodes to a prefix)
            assert data.startswith(got)
        except RsnError as e:
            assert e.code == -3


def test_pipelined_lzss_compress_is_the_serial_compress(oracle):
    """rsn_lzss_compress from 128 MiB up: sections of a quarter of the input, encoded as they land (lzss_encode_sliced) -- the serial call's
    bytes (RSN_HOST_SERIAL=1, a process of its own) on text, on text with '<' in it (mapped to FF where a section is loaded), on a short
    period and on mixed sections; an input with a byte that needs an escape, early or late, is encoded whole: the same bytes again."""
    import workloads as W
    from raisin_amd import lz
    n = 136 << 20
    text = bytes(W.config_input("4", n).numpy())
    lt = bytearray(text); lt[7::1000] = b"<" * len(lt[7::1000]); lt = bytes(lt)
    per = (bytes(range(33, 127)) * 50)[:4096 - 17]
    mixed = text[: 40 << 20] + (per * ((50 << 20) // len(per) + 1))[: 50 << 20] + bytes(30 << 20) + text[40 << 20: 56 << 20]
    esc_early = b"\\" + text[1:]
    esc_late = text[:-5] + b"\xff" + text[-4:]
    datas = {"text": text, "lt": lt, "mixed": mixed, "esc_early": esc_early, "esc_late": esc_late, "odd length": text[3:-2]}
    got = {k: hashlib.sha256(lz.CompressAsync(v)).hexdigest() for k, v in datas.items()}
    code = ("import sys, hashlib, pickle; sys.path.insert(0, %r)\nfrom raisin_amd import lz\n"
            "datas = pickle.load(open(sys.argv[1], 'rb'))\n"
            "print(' '.join(k + ':' + hashlib.sha256(lz.CompressAsync(v)).hexdigest() for k, v in sorted(datas.items())))\n" % ROOT)
    import pickle
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".pkl") as f:
        pickle.dump(datas, f)
        f.flush()
        out = subprocess.run([sys.executable, "-c", code, f.name], capture_output=True, text=True, timeout=900, env=dict(os.environ, RSN_HOST_SERIAL="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == " ".join(k + ":" + v for k, v in sorted(got.items()))
    c = lz.CompressAsync(lt)
    assert lz.Decompress(c) == lt
    assert c[: 1 << 20] == oracle.lzss_compress(lt[: 3 << 20])[: 1 << 20]        # (and the oracle's, where it can be had in seconds)


def test_pipelined_lzss_decompress_is_the_serial_decompress(oracle):
    """rsn_lzss_decompress from 64 MiB up (r06): the stream decoded slice by slice as it lands (lzss_decode_sliced), the result block sized
    from a host-side parse of the first 4 MiB.  The serial call's bytes (RSN_HOST_SERIAL=1, a process of its own) on text's stream, on a
    stream whose expansion GROWS behind the sample (text, then a period: the slices outgrow the block -- the serial call answers), on one that
    shrinks, on streams with a 5C early and late (escapes: the serial call), on a hand-made stream whose tokens straddle the slice cuts, and a
    malformed token late in the stream (an error either way)."""
    import pickle
    import tempfile
    import workloads as W
    from raisin_amd import RsnError, lz
    n = 150 << 20
    text = bytes(W.config_input("4", n).numpy())
    ctext = lz.CompressAsync(text)
    per = bytes(range(33, 127)) * 44
    grow = lz.CompressAsync(text[: 100 << 20] + (per * ((400 << 20) // len(per) + 1))[: 400 << 20])
    shrink = lz.CompressAsync((per * ((100 << 20) // len(per) + 1))[: 100 << 20] + text)
    esc_early = lz.CompressAsync(b"\\" + text[1:])
    esc_late = lz.CompressAsync(text[:-5] + b"\xff" + text[-4:])
    unit = b"0123456789abcdefghijklmnopqrstuvwxyz" * 3 + b"<108,108>" + b"<7,7>" + b"Q"          # tokens at every alignment against the 4 KiB blocks and the 64 MiB cuts
    straddle = unit * ((70 << 20) // len(unit))
    streams = {"text": ctext, "grow": grow, "shrink": shrink, "esc_early": esc_early, "esc_late": esc_late, "straddle": straddle}
    assert all(len(v) >= (64 << 20) for v in streams.values()), {k: len(v) >> 20 for k, v in streams.items()}
    got = {k: hashlib.sha256(lz.Decompress(v)).hexdigest() for k, v in streams.items()}
    assert lz.Decompress(ctext) == text
    code = ("import sys, hashlib, pickle; sys.path.insert(0, %r)\nfrom raisin_amd import lz\n"
            "streams = pickle.load(open(sys.argv[1], 'rb'))\n"
            "print(' '.join(k + ':' + hashlib.sha256(lz.Decompress(v)).hexdigest() for k, v in sorted(streams.items())))\n" % ROOT)
    with tempfile.NamedTemporaryFile(suffix=".pkl") as f:
        pickle.dump(streams, f)
        f.flush()
        out = subprocess.run([sys.executable, "-c", code, f.name], capture_output=True, text=True, timeout=900, env=dict(os.environ, RSN_HOST_SERIAL="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == " ".join(k + ":" + v for k, v in sorted(got.items()))
    bad = ctext[: 100 << 20] + b"<12,x>" + ctext[100 << 20:]
    with pytest.raises(RsnError) as e:
        lz.Decompress(bad)
    assert e.value.code == -3


def test_four_pipelined_calls_at_once():
    """Four threads, each a pipelined host call at the same moment (LZSS decode and Huffman decode, 70 MiB streams): every call draws its
    two helpers from the library's pool (eight at once), and a second round finds them there -- the results are the single calls'."""
    import threading
    from raisin_amd import huffman, lz
    unit = b"0123456789abcdefghijklmnopqrstuvwxyz" * 3 + b"<108,108>" + b"<7,7>" + b"Q"
    lzs = unit * ((70 << 20) // len(unit))
    rng = np.random.default_rng(77)
    hin = rng.integers(0, 128, size=80 << 20, dtype=np.uint8).tobytes()
    hs = huffman.Compress(hin)
    want_l = hashlib.sha256(lz.Decompress(lzs)).hexdigest()
    assert huffman.Decompress(hs) == hin
    errors = []

    def work(t):
        try:
            for r in range(2):
                if (t + r) % 2:
                    if hashlib.sha256(lz.Decompress(lzs)).hexdigest() != want_l:
                        errors.append(("lzss", t, r))
                elif huffman.Decompress(hs) != hin:
                    errors.append(("huffman", t, r))
        except Exception as e:          # noqa: BLE001
            errors.append((t, repr(e)))
    ts = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_two_pipelined_calls_on_overlapping_views_of_one_buffer():
    """Two goroutines may compress overlapping slices of ONE buffer at the same moment (the input is borrowed and read-only): the second
    call's pieces overlap what the first has registered without being the same ranges -- held, not registered again, copied stretch by
    stretch; nobody's registration is pulled away under the other's copies (r06, the table in rsn_api.hip)."""
    import ctypes
    import threading
    import workloads as W
    from raisin_amd import _lib
    L = _lib.lib()
    buf = W.config_input("4", 150 << 20).numpy()
    views = [buf, buf[(3 << 20) + 17:], buf[: 140 << 20]]

    def call(a):
        out = ctypes.POINTER(ctypes.c_uint8)()
        got = ctypes.c_size_t(0)
        _lib.check(L.rsn_lzss_compress(a.ctypes.data_as(ctypes.c_char_p), a.size, 4096, ctypes.byref(out), ctypes.byref(got)))
        h = hashlib.sha256(ctypes.string_at(out, got.value)).hexdigest()
        L.rsn_free(out)
        return h
    want = [call(v) for v in views]

    def hcall(a):                                                   # the engine's other goroutine: ANOTHER codec on the same bytes (engine.go:235-244), a serial upload
        out = ctypes.POINTER(ctypes.c_uint8)()
        got = ctypes.c_size_t(0)
        _lib.check(L.rsn_huffman_compress(a.ctypes.data_as(ctypes.c_char_p), a.size, ctypes.byref(out), ctypes.byref(got)))
        h = hashlib.sha256(ctypes.string_at(out, got.value)).hexdigest()
        L.rsn_free(out)
        return h
    views.append(buf)
    want.append(hcall(buf))
    errors = []

    def work(t):
        try:
            for r in range(2):
                if (hcall(views[t]) if t == 3 else call(views[t])) != want[t]:
                    errors.append((t, r))
        except Exception as e:          # noqa: BLE001
            errors.append((t, repr(e)))
    ts = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
