//go:build rsn

// Overlay for go-compression/raisin compressor/huffman: Compress/Decompress backed by
// librsn (include/rsn.h).  Drop next to huffman.go and give the pure-Go Compress/Decompress
// the tag `//go:build !rsn`.  Written without a Go toolchain (none exists in the build
// image): see INTEGRATION.md.
package huffman

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
	runtime.LockOSThread() // librsn keeps its device context per OS thread
	defer runtime.UnlockOSThread()
	var p *C.uint8_t
	if len(in) > 0 {
		p = (*C.uint8_t)(unsafe.Pointer(&in[0]))
	}
	var out *C.uint8_t
	var n C.size_t
	if rc := f(p, C.size_t(len(in)), &out, &n); rc != 0 {
		panic("librsn: " + C.GoString(C.rsn_last_error())) // the reference panics via check(e)
	}
	defer C.rsn_free(unsafe.Pointer(out))
	return C.GoBytes(unsafe.Pointer(out), C.int(n))
}

// Compress replaces huffman.go:299.
func Compress(fileContents []byte) []byte {
	return rsnCall(fileContents, func(p *C.uint8_t, n C.size_t, o **C.uint8_t, on *C.size_t) C.int {
		return C.rsn_huffman_compress(p, n, o, on)
	})
}

// Decompress replaces huffman.go:327.
func Decompress(fileContents []byte) []byte {
	return rsnCall(fileContents, func(p *C.uint8_t, n C.size_t, o **C.uint8_t, on *C.size_t) C.int {
		return C.rsn_huffman_decompress(p, n, o, on)
	})
}
