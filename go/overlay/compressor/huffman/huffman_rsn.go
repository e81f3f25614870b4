//go:build rsn
// +build rsn

// Overlay for go-compression/raisin compressor/huffman: Compress/Decompress backed by librsn
// (include/rsn.h).  Build tags select whole FILES, and huffman.go also defines Writer, Reader,
// NewWriter, NewReader and the helpers the engine type-asserts (engine.go:69,121) -- so huffman.go
// itself must stay untagged.  Integration is therefore two steps (INTEGRATION.md):
//   1. move the bodies of Compress (huffman.go:299-325) and Decompress (:327-330) -- and nothing
//      else -- out of huffman.go into a new huffman_purego.go that starts with `//go:build !rsn`;
//   2. drop this file next to it.
// Written without a Go toolchain (none exists in the build image); tests/abi_shim_test.c replays
// this file's exact call sequence against librsn in C.
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
		p = (*C.uint8_t)(unsafe.Pointer(&in[0])) // borrowed for the call only; cgo pins it
	}
	var out *C.uint8_t
	var n C.size_t
	if rc := f(p, C.size_t(len(in)), &out, &n); rc != 0 {
		panic("librsn: " + C.GoString(C.rsn_last_error())) // the reference panics via check(e)
	}
	defer C.rsn_free(unsafe.Pointer(out))
	// Not C.GoBytes (its length is a C.int: results reach 2 GiB and more, librsn takes 5 GiB Huffman calls) and not
	// unsafe.Slice (Go >= 1.17; the reference's go.mod says `go 1.15`, under which it is a compile error whatever the
	// toolchain): the C block is viewed through the classic array-pointer conversion, at most 1 GiB at a time.
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
