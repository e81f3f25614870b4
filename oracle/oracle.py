"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (raisin_amd) never imports it.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("huffman_oracle.c", "lzss_oracle.c", "cpu_baseline.c", "rsn_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _SO


class OracleError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        outp = ctypes.POINTER(u8p)
        szp = ctypes.POINTER(ctypes.c_size_t)
        for name, extra in (
            ("rsn_oracle_huffman_compress", []),
            ("rsn_oracle_lzss_decompress", []),
            ("rsn_oracle_lzss_escape", []),
            ("rsn_oracle_lzss_unescape", []),
        ):
            f = getattr(L, name)
            f.argtypes = [ctypes.c_char_p, ctypes.c_size_t] + extra + [outp, szp]
            f.restype = ctypes.c_int
        L.rsn_oracle_huffman_decompress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, outp, szp]
        L.rsn_oracle_huffman_decompress.restype = ctypes.c_int
        for name in ("rsn_oracle_lzss_compress", "rsn_oracle_lzss_compress_allpos", "rsn_oracle_lzss_compress_legacy"):
            f = getattr(L, name)
            f.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int64, outp, szp]
            f.restype = ctypes.c_int
        L.rsn_oracle_free.argtypes = [ctypes.c_void_p]
        L.rsn_oracle_last_error.restype = ctypes.c_char_p
        L.rsn_oracle_huffman_table.restype = ctypes.c_int64
        L.rsn_oracle_huffman_table.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.rsn_oracle_utf8_runes.restype = ctypes.c_size_t
        L.rsn_oracle_utf8_runes.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
        L.rsn_oracle_lzss_matches.restype = ctypes.c_int
        for name in ("rsn_baseline_huffman_compress_mt", "rsn_baseline_huffman_decompress_mt"):
            f = getattr(L, name)
            f.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, outp, szp]
            f.restype = ctypes.c_int
        L.rsn_baseline_lzss_compress_mt.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t, outp, szp]
        L.rsn_baseline_lzss_compress_mt.restype = ctypes.c_int
        L.rsn_baseline_lzss_check.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t,
                                              ctypes.c_char_p, ctypes.c_size_t, szp]
        L.rsn_baseline_lzss_check.restype = ctypes.c_int
        L.rsn_oracle_lzss_matches.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        _lib = L
    return _lib


def _call(fn, data, *extra):
    L = lib()
    data = bytes(data)
    out = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_size_t(0)
    rc = fn(data, len(data), *extra, ctypes.byref(out), ctypes.byref(n))
    if rc != 0:
        raise OracleError(L.rsn_oracle_last_error().decode("utf-8", "replace"))
    try:
        return ctypes.string_at(out, n.value)
    finally:
        L.rsn_oracle_free(out)


def huffman_compress(data):
    return _call(lib().rsn_oracle_huffman_compress, data)


def huffman_decompress(data, strict_ref_limit=False):
    return _call(lib().rsn_oracle_huffman_decompress, data, int(strict_ref_limit))


def lzss_compress(data, window=4096):
    return _call(lib().rsn_oracle_lzss_compress, data, window)


def lzss_compress_allpos(data, window=4096):
    return _call(lib().rsn_oracle_lzss_compress_allpos, data, window)


def lzss_compress_legacy(data, window=4096):
    return _call(lib().rsn_oracle_lzss_compress_legacy, data, window)


def lzss_decompress(data):
    return _call(lib().rsn_oracle_lzss_decompress, data)


def lzss_escape(data):
    return _call(lib().rsn_oracle_lzss_escape, data)


def lzss_unescape(data):
    return _call(lib().rsn_oracle_lzss_unescape, data)


# ---- the threaded CPU baseline (oracle/cpu_baseline.c): same bytes as the functions above, on `threads` host cores
def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def huffman_compress_mt(data, threads):
    return _call(lib().rsn_baseline_huffman_compress_mt, data, int(threads))


def huffman_decompress_mt(data, threads):
    return _call(lib().rsn_baseline_huffman_decompress_mt, data, int(threads))


def lzss_compress_mt(data, window=4096, threads=1, grain=4096):
    """grain = positions per task; 1 = one task per position, the reference's goroutine-per-byte shape (lzss.go:117-130)."""
    return _call(lib().rsn_baseline_lzss_compress_mt, data, window, int(threads), int(grain))


def lzss_check(data, candidate, window=4096, threads=None, seg=1 << 18):
    """True iff `candidate` is byte for byte lzss_compress(data, window) -- decided by induction over segments cut after the
    candidate's tokens, each re-encoded by the oracle's own greedy loop on its own thread (oracle/cpu_baseline.c): what makes
    an oracle-exact comparison affordable at 1 GiB.  Returns (ok, offset of the first differing segment in `candidate`)."""
    bad = ctypes.c_size_t(0)
    data, candidate = bytes(data), bytes(candidate)
    rc = lib().rsn_baseline_lzss_check(data, len(data), window, int(threads or host_cores()), int(seg), candidate, len(candidate), ctypes.byref(bad))
    if rc < 0:
        raise OracleError(lib().rsn_oracle_last_error().decode("utf-8", "replace"))
    return rc == 0, bad.value


def lzss_matches(escaped, window=4096):
    import numpy as np
    escaped = bytes(escaped)
    off = np.zeros(len(escaped), dtype=np.uint32)
    size = np.zeros(len(escaped), dtype=np.uint32)
    lib().rsn_oracle_lzss_matches(escaped, len(escaped), window, off.ctypes.data, size.ctypes.data)
    return off, size


def utf8_runes(data):
    import numpy as np
    data = bytes(data)
    r = np.zeros(max(len(data), 1), dtype=np.uint32)
    k = lib().rsn_oracle_utf8_runes(data, len(data), r.ctypes.data)
    return r[:k]


def huffman_table(data):
    """[(rune, freq, code, len)] in printCodes DFS order (huffman.go:110)."""
    import numpy as np
    data = bytes(data)
    cap = 0x110000
    runes = np.zeros(cap, dtype=np.uint32)
    freqs = np.zeros(cap, dtype=np.uint64)
    codes = np.zeros(cap, dtype=np.uint64)
    lens = np.zeros(cap, dtype=np.uint8)
    a = lib().rsn_oracle_huffman_table(data, len(data), runes.ctypes.data, freqs.ctypes.data, codes.ctypes.data,
                                       lens.ctypes.data, cap)
    if a < 0:
        raise OracleError(lib().rsn_oracle_last_error().decode())
    return [(int(runes[i]), int(freqs[i]), int(codes[i]), int(lens[i])) for i in range(a)]


def header_entries(rsn):
    """Split a huffman .rsn header into the multiset of `freq|symbol` entries
    (order-independent comparison: the reference's order is Go map order,
    huffman.go:312)."""
    sep = rsn.index(b"\\\n")
    h = rsn[:sep]
    out = []
    i = 0
    while i < len(h):
        j = h.index(b"|", i)
        freq = h[i:j]
        k = j + 1
        if h[k:k + 2] == b"\\n":
            k += 2
        else:
            b0 = h[k]
            k += 1
            while k < len(h) and 0x80 <= h[k] <= 0xBF and b0 >= 0xC0:
                k += 1
        out.append((freq, h[j + 1:k]))
        i = k
    return sorted(out), rsn[sep + 2:]
