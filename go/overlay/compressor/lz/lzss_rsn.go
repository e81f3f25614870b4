//go:build rsn
// +build rsn

// Overlay for go-compression/raisin compressor/lz: CompressAsync / Compress / Decompress backed by
// librsn (include/rsn.h).  lzss.go also defines Writer, Reader, NewWriter, NewWriterLevel,
// NewReader, EncodeOpeningSymbols, ... and must stay untagged; integration is two steps
// (INTEGRATION.md):
//   1. go/overlay/split.py moves CompressAsync (lzss.go:109-154), compressorWorkerAsync (:156-164),
//      Compress (:224-316) and Decompress (:323-364) with the `sync` import out of lzss.go into a
//      new lzss_purego.go (`//go:build !rsn`); compressorWorker (:166-184) stays, CompressRecursive
//      (:203) calls it;
//   2. and drops this file next to it.
// Go packages cannot share unexported helpers, so rsnCall is repeated from the huffman overlay.
// Written without a Go toolchain; tests/abi_shim_test.c replays this call sequence in C.
package lz

/*
#cgo CFLAGS: -I${SRCDIR}/../../third_party/librsn/include
#cgo LDFLAGS: -L${SRCDIR}/../../third_party/librsn -lrsn -Wl,-rpath,${SRCDIR}/../../third_party/librsn
#include <stdlib.h>
#include "rsn.h"
*/
import "C"

import (
	"runtime"
	"unsafe"
)

func rsnCall(in []byte, f func(p *C.uint8_t, n C.size_t, out **C.uint8_t, outN *C.size_t) C.int) []byte {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	var p *C.uint8_t
	if len(in) > 0 {
		p = (*C.uint8_t)(unsafe.Pointer(&in[0]))
	}
	var out *C.uint8_t
	var n C.size_t
	if rc := f(p, C.size_t(len(in)), &out, &n); rc != 0 {
		panic("librsn: " + C.GoString(C.rsn_last_error()))
	}
	defer C.rsn_free(unsafe.Pointer(out))
	// Not C.GoBytes (its length is a C.int: results reach 2 GiB and more) and not unsafe.Slice (Go >= 1.17; the
	// reference's go.mod says `go 1.15`, under which it is a compile error whatever the toolchain): the C block is
	// viewed through the classic array-pointer conversion, at most 1 GiB at a time.
	res := make([]byte, int(n))
	const view = 1 << 30
	for off := 0; off < int(n); off += view {
		m := int(n) - off
		if m > view {
			m = view
		}
		src := (*[view]byte)(unsafe.Pointer(uintptr(unsafe.Pointer(out)) + uintptr(off)))[:m:m]
		copy(res[off:], src)
	}
	return res
}

// CompressAsync replaces lzss.go:109 (the engine path, Writer.Write lzss.go:53-57).
// The progress bar (lzss.go:113-115) has no equivalent.
func CompressAsync(fileContents []byte, useProgressBar bool, maxSearchBufferLength int) []byte {
	return rsnCall(fileContents, func(p *C.uint8_t, n C.size_t, o **C.uint8_t, on *C.size_t) C.int {
		return C.rsn_lzss_compress(p, n, C.int64_t(maxSearchBufferLength), o, on)
	})
}

// Compress replaces lzss.go:224, the older synchronous encoder (host code in librsn, quirks kept).
func Compress(fileContents []byte, useProgressBar bool, maxSearchBufferLength int) []byte {
	return rsnCall(fileContents, func(p *C.uint8_t, n C.size_t, o **C.uint8_t, on *C.size_t) C.int {
		return C.rsn_lzss_compress_legacy(p, n, C.int64_t(maxSearchBufferLength), o, on)
	})
}

// Decompress replaces lzss.go:323.
func Decompress(fileContents []byte, useProgressBar bool) []byte {
	return rsnCall(fileContents, func(p *C.uint8_t, n C.size_t, o **C.uint8_t, on *C.size_t) C.int {
		return C.rsn_lzss_decompress(p, n, o, on)
	})
}
