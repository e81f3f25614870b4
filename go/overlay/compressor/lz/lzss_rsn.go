//go:build rsn

// Overlay for go-compression/raisin compressor/lz: CompressAsync/Decompress backed by librsn
// (include/rsn.h).  Drop next to lzss.go and give the pure-Go CompressAsync/Decompress the
// tag `//go:build !rsn`.  Written without a Go toolchain: see INTEGRATION.md.
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
	return C.GoBytes(unsafe.Pointer(out), C.int(n))
}

// CompressAsync replaces lzss.go:109 (the engine path, Writer.Write lzss.go:53-57).
// The progress bar (lzss.go:113-115) has no equivalent.
func CompressAsync(fileContents []byte, useProgressBar bool, maxSearchBufferLength int) []byte {
	return rsnCall(fileContents, func(p *C.uint8_t, n C.size_t, o **C.uint8_t, on *C.size_t) C.int {
		return C.rsn_lzss_compress(p, n, C.int64_t(maxSearchBufferLength), o, on)
	})
}

// Decompress replaces lzss.go:323.
func Decompress(fileContents []byte, useProgressBar bool) []byte {
	return rsnCall(fileContents, func(p *C.uint8_t, n C.size_t, o **C.uint8_t, on *C.size_t) C.int {
		return C.rsn_lzss_decompress(p, n, o, on)
	})
}
