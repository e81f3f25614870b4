"""ctypes binding of librsn.so (include/rsn.h).  The library is the product: if
it is missing or cannot reach a HIP device every call raises -- there is no
Python or CPU fallback anywhere in this package."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSN_LIB_PATH") or os.path.join(_HERE, "librsn.so")   # RSN_LIB_PATH: another BUILD of librsn (A/B of compile-time switches)

RSN_OK = 0
RSN_ERR_CAPACITY = -7


class RsnError(RuntimeError):
    """Raised where the reference would panic (check(e) / index out of range)."""

    def __init__(self, code, msg):
        super().__init__("librsn error %d: %s" % (code, msg))
        self.code = code


class ProfEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("launches", ctypes.c_uint64), ("total_ms", ctypes.c_double)]


_lib = None

SYMBOLS = [
    "rsn_device_set", "rsn_device_count", "rsn_last_error", "rsn_version", "rsn_free", "rsn_trim",
    "rsn_huffman_compress", "rsn_huffman_decompress", "rsn_lzss_compress", "rsn_lzss_decompress", "rsn_lzss_compress_legacy",
    "rsn_huffman_compress_batch", "rsn_huffman_compress_sharded",
    "rsn_huffman_compress_bound", "rsn_lzss_compress_bound",
    "rsn_huffman_compress_dev", "rsn_huffman_decompress_dev", "rsn_lzss_compress_dev", "rsn_lzss_decompress_dev",
    "rsn_prof_enable", "rsn_prof_reset", "rsn_prof_get", "rsn_huffman_table",
    "rsn_huffman_plan", "rsn_huffman_parse_header", "rsn_huffman_slice_cuts",
]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("raisin_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C raisin_amd/csrc`; there is no fallback path" % LIB_PATH)
    # librsn links the system ROCm runtime (/opt/rocm); PyTorch bundles its own
    # copy.  Both can live in one process (device memory is shared by address),
    # but only if PyTorch's copy initialises first -- so import it here when it
    # is installed.  Stream/event HANDLES are never exchanged between the two
    # runtimes: librsn always runs on its own stream (see tensor helpers).
    try:
        import torch  # noqa: F401
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    u8p = ctypes.POINTER(ctypes.c_uint8)
    vp, sz, szp = ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)
    L.rsn_last_error.restype = ctypes.c_char_p
    L.rsn_version.restype = ctypes.c_char_p
    L.rsn_trim.argtypes = []
    L.rsn_trim.restype = None
    L.rsn_free.argtypes = [vp]
    L.rsn_free.restype = None
    L.rsn_device_set.argtypes = [ctypes.c_int]
    for name in ("rsn_huffman_compress", "rsn_huffman_decompress", "rsn_lzss_decompress"):
        getattr(L, name).argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(u8p), szp]
    L.rsn_lzss_compress.argtypes = [ctypes.c_char_p, sz, ctypes.c_int64, ctypes.POINTER(u8p), szp]
    L.rsn_lzss_compress_legacy.argtypes = [ctypes.c_char_p, sz, ctypes.c_int64, ctypes.POINTER(u8p), szp]
    L.rsn_huffman_compress_bound.argtypes = [sz]
    L.rsn_huffman_compress_bound.restype = sz
    L.rsn_lzss_compress_bound.argtypes = [sz]
    L.rsn_lzss_compress_bound.restype = sz
    for name in ("rsn_huffman_compress_dev", "rsn_huffman_decompress_dev", "rsn_lzss_decompress_dev"):
        getattr(L, name).argtypes = [vp, sz, vp, sz, szp, vp]
    L.rsn_lzss_compress_dev.argtypes = [vp, sz, ctypes.c_int64, vp, sz, szp, vp]
    L.rsn_prof_enable.argtypes = [ctypes.c_int]
    L.rsn_prof_enable.restype = None
    L.rsn_prof_reset.restype = None
    L.rsn_prof_get.argtypes = [ctypes.POINTER(ProfEntry), ctypes.c_int]
    L.rsn_huffman_table.argtypes = [ctypes.c_char_p, sz, vp, vp, vp, vp, sz]
    L.rsn_huffman_table.restype = ctypes.c_int64
    L.rsn_huffman_plan.argtypes = [vp, vp, sz, vp, vp, vp, vp, sz, szp]
    L.rsn_huffman_plan.restype = ctypes.c_int64
    L.rsn_huffman_parse_header.argtypes = [ctypes.c_char_p, sz, vp, vp, sz]
    L.rsn_huffman_parse_header.restype = ctypes.c_int64
    L.rsn_huffman_slice_cuts.argtypes = [ctypes.c_char_p, sz, ctypes.c_int, szp, sz]
    L.rsn_huffman_slice_cuts.restype = ctypes.c_int64
    L.rsn_huffman_compress_sharded.argtypes = [ctypes.c_char_p, sz, ctypes.c_int, ctypes.POINTER(u8p), szp]
    L.rsn_huffman_compress_batch.argtypes = [sz, ctypes.POINTER(ctypes.c_char_p), szp, ctypes.POINTER(u8p), szp]
    _lib = L
    return L


def check(rc):
    if rc != RSN_OK:
        raise RsnError(rc, lib().rsn_last_error().decode("utf-8", "replace"))


def call_host(fn, data, *extra):
    """bytes in -> bytes out through a host-buffer entry point."""
    L = lib()
    data = bytes(data)
    out = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_size_t(0)
    check(fn(data, len(data), *extra, ctypes.byref(out), ctypes.byref(n)))
    try:
        return ctypes.string_at(out, n.value)
    finally:
        L.rsn_free(out)


def call_dev(fn, d_in, n, d_out, cap, stream, *extra):
    """device pointers in/out; returns the produced size.  On RSN_ERR_CAPACITY
    raises RsnError whose .needed holds the capacity that would have sufficed."""
    got = ctypes.c_size_t(0)
    rc = fn(d_in, n, *extra, d_out, cap, ctypes.byref(got), stream)
    if rc != RSN_OK:
        err = RsnError(rc, lib().rsn_last_error().decode("utf-8", "replace"))
        err.needed = got.value
        raise err
    return got.value


def prof_enable(on=True):
    lib().rsn_prof_enable(1 if on else 0)


def prof_reset():
    lib().rsn_prof_reset()


def prof_get():
    L = lib()
    arr = (ProfEntry * 64)()
    k = L.rsn_prof_get(arr, 64)
    return {arr[i].name.decode(): (int(arr[i].launches), float(arr[i].total_ms)) for i in range(min(k, 64))}


def own_stream(tensor, stream=None):
    """Orders a librsn call after the work already queued on the tensor's torch
    stream and returns the stream argument for the C ABI.  `stream` must be a
    hipStream_t created by the SAME HIP runtime librsn links (never a torch
    stream handle: PyTorch carries its own runtime copy); None = librsn's own
    per-thread stream."""
    import torch
    torch.cuda.current_stream(tensor.device).synchronize()
    lib().rsn_device_set(tensor.device.index or 0)
    return stream
