"""Mirror of /root/reference/compressor/huffman (huffman.go) over librsn."""
import io

from . import _lib


def Compress(fileContents):
    """huffman.go:299 Compress([]byte) []byte"""
    return _lib.call_host(_lib.lib().rsn_huffman_compress, fileContents)


def Decompress(fileContents):
    """huffman.go:327 Decompress([]byte) []byte"""
    return _lib.call_host(_lib.lib().rsn_huffman_decompress, fileContents)


def CompressSharded(fileContents, shards=0):
    """ONE stream from `shards` slices of one input (rsn_huffman_compress_sharded): per-slice histograms summed, one tree, every slice
    encoded at its bit offset by a worker of its own (a device of its own with RSN_BATCH_DEVICES).  Equals Compress(fileContents)."""
    return _lib.call_host(_lib.lib().rsn_huffman_compress_sharded, fileContents, int(shards))


def CompressBatch(chunks):
    """One complete .rsn segment per chunk, as engine.CompressFiles writes one file per input (engine.go:150-154):
    rsn_huffman_compress_batch deals the chunks out over the visible GPUs (chunk k -> device k mod G) and pipelines
    upload / encode / download per device.  Each result equals Compress(chunk)."""
    import ctypes
    L = _lib.lib()
    chunks = [bytes(c) for c in chunks]
    k = len(chunks)
    ins = (ctypes.c_char_p * k)(*chunks)
    lens = (ctypes.c_size_t * k)(*[len(c) for c in chunks])
    outs = (ctypes.POINTER(ctypes.c_uint8) * k)()
    olens = (ctypes.c_size_t * k)()
    _lib.check(L.rsn_huffman_compress_batch(k, ins, lens, outs, olens))
    try:
        return [ctypes.string_at(outs[i], olens[i]) for i in range(k)]
    finally:
        for i in range(k):
            L.rsn_free(outs[i])


class Writer:
    """huffman.go:368-386: Write compresses the whole buffer once and returns len(compressed)."""

    def __init__(self, w):
        self.w = w

    def Write(self, data):
        compressed = Compress(data)
        self.w.write(compressed)
        return len(compressed)

    write = Write

    def Close(self):
        return None

    close = Close


class Reader:
    """huffman.go:388-422: the first Read drains the source and decompresses everything."""

    def __init__(self, r):
        self.r = r
        self.decompressed = None
        self.pos = 0

    def Read(self, size=-1):
        if self.decompressed is None:
            self.decompressed = Decompress(self.r.read())
        if size is None or size < 0:
            size = len(self.decompressed) - self.pos
        chunk = self.decompressed[self.pos:self.pos + size]
        self.pos += len(chunk)
        return chunk

    read = Read


def NewWriter(w):
    """huffman.go:372 NewWriter(io.Writer) io.WriteCloser"""
    return Writer(w)


def NewReader(r):
    """huffman.go:395 NewReader(io.Reader) io.Reader"""
    if isinstance(r, (bytes, bytearray)):
        r = io.BytesIO(r)
    return Reader(r)


def table(data):
    """[(rune, freq, code, len)] in printCodes order (huffman.go:110) as built by the library."""
    import ctypes

    import numpy as np
    L = _lib.lib()
    data = bytes(data)
    cap = 0x110000
    runes = np.zeros(cap, dtype=np.uint32)
    freqs = np.zeros(cap, dtype=np.uint64)
    codes = np.zeros(cap, dtype=np.uint64)
    lens = np.zeros(cap, dtype=np.uint8)
    a = L.rsn_huffman_table(data, len(data), runes.ctypes.data, freqs.ctypes.data, codes.ctypes.data, lens.ctypes.data, cap)
    if a < 0:
        _lib.check(int(a))
    return [(int(runes[i]), int(freqs[i]), int(codes[i]), int(lens[i])) for i in range(a)]


def plan(counts):
    """Host-only: {rune: count} -> ([(rune, code, len)] in printCodes order, header bytes).
    Runs the library's Go-exact tree builder without touching a device."""
    import ctypes

    import numpy as np
    L = _lib.lib()
    items = sorted(counts.items())
    runes = np.array([r for r, _ in items], dtype=np.uint32)
    cnts = np.array([c for _, c in items], dtype=np.uint64)
    n = len(items)
    o_r = np.zeros(max(n, 1), dtype=np.uint32)
    o_c = np.zeros(max(n, 1), dtype=np.uint64)
    o_l = np.zeros(max(n, 1), dtype=np.uint8)
    hdr = np.zeros(32 * max(n, 1), dtype=np.uint8)
    hl = ctypes.c_size_t(0)
    a = L.rsn_huffman_plan(runes.ctypes.data, cnts.ctypes.data, n, o_r.ctypes.data, o_c.ctypes.data, o_l.ctypes.data,
                           hdr.ctypes.data, hdr.size, ctypes.byref(hl))
    if a < 0:
        _lib.check(int(a))
    return [(int(o_r[i]), int(o_c[i]), int(o_l[i])) for i in range(a)], bytes(hdr[:hl.value])


def parse_header(header):
    """Host-only: decodeTree's header scan (huffman.go:196-227) -> [(rune, count)] ascending."""
    import numpy as np
    L = _lib.lib()
    header = bytes(header)
    cap = max(len(header), 1)
    runes = np.zeros(cap, dtype=np.uint32)
    cnts = np.zeros(cap, dtype=np.uint64)
    a = L.rsn_huffman_parse_header(header, len(header), runes.ctypes.data, cnts.ctypes.data, cap)
    if a < 0:
        _lib.check(int(a))
    return [(int(runes[i]), int(cnts[i])) for i in range(a)]


# ---- device-resident form (torch tensors as plain device memory) -----------
from ._lib import own_stream as _own_stream  # noqa: E402

def compress_bound(n):
    return int(_lib.lib().rsn_huffman_compress_bound(n))


def compress_tensor(src, out=None, stream=None):
    """src: uint8 CUDA tensor.  Returns a uint8 tensor holding the .rsn bytes: a view of `out` when it was large
    enough, otherwise (RSN_ERR_CAPACITY) a view of a fresh tensor of the capacity the library asked for."""
    import torch
    n = src.numel()
    if out is None:
        out = torch.empty(n + n // 8 + (1 << 16), dtype=torch.uint8, device=src.device)
    st = _own_stream(src, stream)
    try:
        got = _lib.call_dev(_lib.lib().rsn_huffman_compress_dev, src.data_ptr(), n, out.data_ptr(), out.numel(), st)
    except _lib.RsnError as e:
        if e.code != _lib.RSN_ERR_CAPACITY:
            raise
        out = torch.empty(e.needed, dtype=torch.uint8, device=src.device)
        got = _lib.call_dev(_lib.lib().rsn_huffman_compress_dev, src.data_ptr(), n, out.data_ptr(), out.numel(), st)
    return out[:got]


def decompress_tensor(src, out=None, stream=None):
    import torch
    n = src.numel()
    st = _own_stream(src, stream)
    if out is None and n < (1 << 20):
        out = torch.empty(8 * n + (1 << 16), dtype=torch.uint8, device=src.device)   # small: a generous guess costs less than a second call
    if out is None:
        # the size query first (d_out NULL): a guess of the expansion would be a buffer of many times the input, held by the view returned
        try:
            need = _lib.call_dev(_lib.lib().rsn_huffman_decompress_dev, src.data_ptr(), n, None, 0, st)
        except _lib.RsnError as e:
            if e.code != _lib.RSN_ERR_CAPACITY:
                raise
            need = e.needed
        out = torch.empty(max(need, 16), dtype=torch.uint8, device=src.device)
    try:
        got = _lib.call_dev(_lib.lib().rsn_huffman_decompress_dev, src.data_ptr(), n, out.data_ptr(), out.numel(), st)
    except _lib.RsnError as e:
        if e.code != _lib.RSN_ERR_CAPACITY:
            raise
        out = torch.empty(e.needed, dtype=torch.uint8, device=src.device)
        got = _lib.call_dev(_lib.lib().rsn_huffman_decompress_dev, src.data_ptr(), n, out.data_ptr(), out.numel(), st)
    return out[:got]
