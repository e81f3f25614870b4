"""raisin_amd -- MI355X (gfx950) implementation of go-compression/raisin's
Huffman and LZSS codecs behind the reference's own package surface.

  raisin_amd.huffman   mirrors compressor/huffman  (Compress, Decompress, NewWriter, NewReader)
  raisin_amd.lz        mirrors compressor/lz       (CompressAsync, Decompress, NewWriter, NewWriterLevel, NewReader)
  raisin_amd.engine    mirrors engine              (CompressedFile, Writers, Readers, layering, BenchmarkFile)

All compute happens in raisin_amd/librsn.so (hand-written HIP kernels, C ABI in
include/rsn.h).  Nothing in this package falls back to the CPU.
"""
from . import _lib  # noqa: F401
from ._lib import RsnError  # noqa: F401
