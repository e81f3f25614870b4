"""Mirror of /root/reference/compressor/lz (lzss.go) over librsn."""
import io

from . import _lib

DefaultWindowSize = 4096  # lzss.go:35


def CompressAsync(fileContents, useProgressBar=False, maxSearchBufferLength=DefaultWindowSize):
    """lzss.go:109 CompressAsync([]byte, bool, int) []byte -- the engine/.rsn path
    (Writer.Write, lzss.go:53-57).  The progress bar has no equivalent."""
    return _lib.call_host(_lib.lib().rsn_lzss_compress, fileContents, int(maxSearchBufferLength))


def Compress(fileContents, useProgressBar=False, maxSearchBufferLength=DefaultWindowSize):
    """lzss.go:224 Compress([]byte, bool, int) []byte -- the older synchronous encoder (not the .rsn
    path; host-side, quirks included: see include/rsn.h)."""
    return _lib.call_host(_lib.lib().rsn_lzss_compress_legacy, fileContents, int(maxSearchBufferLength))


def Decompress(fileContents, useProgressBar=False):
    """lzss.go:323 Decompress([]byte, bool) []byte"""
    return _lib.call_host(_lib.lib().rsn_lzss_decompress, fileContents)


class Writer:
    """lzss.go:29-61"""

    def __init__(self, w, windowSize):
        self.w = w
        self.windowSize = windowSize
        self.useProgressBar = True

    def Write(self, data):
        compressed = CompressAsync(data, self.useProgressBar, self.windowSize)
        self.w.write(compressed)
        return len(compressed)

    write = Write

    def Close(self):
        return None

    close = Close


class Reader:
    """lzss.go:63-106"""

    def __init__(self, r):
        self.r = r
        self.decompressed = None
        self.pos = 0

    def Read(self, size=-1):
        if self.decompressed is None:
            self.decompressed = Decompress(self.r.read(), True)
        if size is None or size < 0:
            size = len(self.decompressed) - self.pos
        chunk = self.decompressed[self.pos:self.pos + size]
        self.pos += len(chunk)
        return chunk

    read = Read

    def Close(self):
        return None


def NewWriterLevel(w, level):
    """lzss.go:42-51: level is the window size; negative levels are an error."""
    if level < 0:
        raise ValueError("lzss: invalid compression level: %d" % level)
    return Writer(w, level)


def NewWriter(w):
    """lzss.go:37 NewWriter(io.Writer) io.WriteCloser (window 4096)"""
    return NewWriterLevel(w, DefaultWindowSize)


def NewReader(r):
    """lzss.go:98 NewReader(io.Reader) io.Reader"""
    if isinstance(r, (bytes, bytearray)):
        r = io.BytesIO(r)
    return Reader(r)


from ._lib import own_stream as _own_stream  # noqa: E402


def compress_bound(n):
    return int(_lib.lib().rsn_lzss_compress_bound(n))


def compress_tensor(src, window=DefaultWindowSize, out=None, stream=None):
    import torch
    n = src.numel()
    if out is None:
        out = torch.empty(compress_bound(n), dtype=torch.uint8, device=src.device)
    st = _own_stream(src, stream)
    got = _lib.call_dev(_lib.lib().rsn_lzss_compress_dev, src.data_ptr(), n, out.data_ptr(), out.numel(), st, int(window))
    return out[:got]


def decompress_tensor(src, out=None, stream=None):
    import torch
    n = src.numel()
    st = _own_stream(src, stream)
    if out is None and n < (1 << 20):
        out = torch.empty(16 * n + (1 << 16), dtype=torch.uint8, device=src.device)   # small: a generous guess costs less than a second call
    if out is None:
        # the size query first (d_out NULL): a guess of the expansion would be a buffer of many times the input, held by the view returned
        try:
            need = _lib.call_dev(_lib.lib().rsn_lzss_decompress_dev, src.data_ptr(), n, None, 0, st)
        except _lib.RsnError as e:
            if e.code != _lib.RSN_ERR_CAPACITY:
                raise
            need = e.needed
        out = torch.empty(max(need, 16), dtype=torch.uint8, device=src.device)
    try:
        got = _lib.call_dev(_lib.lib().rsn_lzss_decompress_dev, src.data_ptr(), n, out.data_ptr(), out.numel(), st)
    except _lib.RsnError as e:
        if e.code != _lib.RSN_ERR_CAPACITY:
            raise
        out = torch.empty(e.needed, dtype=torch.uint8, device=src.device)
        got = _lib.call_dev(_lib.lib().rsn_lzss_decompress_dev, src.data_ptr(), n, out.data_ptr(), out.numel(), st)
    return out[:got]
