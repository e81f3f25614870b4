"""ctypes binding of librsn.so (include/rsn.h).  The library is the product: if
it is missing or cannot reach a HIP device every call raises -- there is no
Python or CPU fallback anywhere in this package."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSN_LIB_PATH") or os.path.join(_HERE, "librsn.so")   # RSN_LIB_PATH: another BUILD of librsn (A/B of compile-time switches)

RSN_OK = 0
RSN_ERR_CAPACITY = -7

# RSN_NO_TORCH=1: this package never imports torch -- bytes, numpy and ctypes only.  librsn then runs on the HIP runtime it is linked
# against (the system's /opt/rocm, ROCm 7.2), which is the runtime a C or Go (cgo) host gets; with torch imported first the process
# resolves libamdhip64.so.7 to the copy torch bundles (ROCm 7.0.2, same SONAME) and librsn runs on THAT one -- the two differ in
# behaviour (under torch's, a download does not start while an upload runs: DESIGN 0, INTEGRATION.md "Which HIP runtime").  The
# parity suites that only need the host-buffer API run once in each mode (tests/test_gpu_no_torch.py).
NO_TORCH = os.environ.get("RSN_NO_TORCH") == "1"


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
    # librsn is LINKED against the system ROCm runtime (/opt/rocm, libamdhip64.so.7); PyTorch bundles a copy of an older release
    # under the same SONAME.  A process holds ONE library per SONAME: whichever is loaded first serves both, so with torch imported
    # first (below: the tensor helpers and bench.py need it, and torch must initialise its own copy) librsn runs on torch's runtime,
    # not on the one it was linked against (hipRuntimeGetVersion 70051831 against 70226015, r05).  Device memory is shared by
    # address either way; stream / event HANDLES are never exchanged (librsn always runs on its own stream, see the tensor helpers).
    # RSN_NO_TORCH=1 skips the import: the runtime a C / Go host gets.
    if not NO_TORCH:
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


_hip = None


def hip():
    """The HIP runtime this process's librsn runs on, through ctypes (the SONAME librsn links: already loaded, so this is the same
    library object) -- device memory without torch: the torch-free mode's way to the device-pointer entry points."""
    global _hip
    if _hip is None:
        lib()
        H = ctypes.CDLL("libamdhip64.so.7")
        H.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        H.hipFree.argtypes = [ctypes.c_void_p]
        H.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        H.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        H.hipRuntimeGetVersion.argtypes = [ctypes.POINTER(ctypes.c_int)]
        _hip = H
    return _hip


def runtime_info():
    """(hipRuntimeGetVersion, path of the libamdhip64 mapped into this process) -- which of the two runtimes librsn is running on."""
    v = ctypes.c_int(0)
    hip().hipRuntimeGetVersion(ctypes.byref(v))
    paths = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln})
    return v.value, paths


def dev_codec(fn, data, cap, *extra):
    """bytes -> device buffer -> a device-pointer entry point (`fn`: rsn_*_dev) -> bytes, without torch: hipMalloc / hipMemcpy of the
    runtime librsn runs on.  The general path for inputs the host-buffer entry points would route to the small-input codec."""
    H, L = hip(), lib()
    data = bytes(data)
    n = len(data)
    check(L.rsn_device_set(0))
    d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()

    def ok(e):
        if e != 0:
            raise RuntimeError("HIP error %d" % e)
    ok(H.hipMalloc(ctypes.byref(d_in), n + 64))
    try:
        ok(H.hipMemset(d_in, 0, n + 64))
        ok(H.hipMemcpy(d_in, data, n, 1))
        for attempt in range(2):
            ok(H.hipMalloc(ctypes.byref(d_out), cap + 64))
            try:
                got = call_dev(fn, d_in, n, d_out, cap, None, *extra)
                buf = ctypes.create_string_buffer(max(got, 1))
                ok(H.hipMemcpy(buf, d_out, got, 2))
                return buf.raw[:got]
            except RsnError as e:
                if e.code != RSN_ERR_CAPACITY or attempt:
                    raise
                cap = e.needed
            finally:
                H.hipFree(d_out)
    finally:
        H.hipFree(d_in)


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
