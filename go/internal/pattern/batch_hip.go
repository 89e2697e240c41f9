//go:build ovalhip

package pattern

/*
#include <stdint.h>
#include <stdlib.h>
#include "oval_pattern.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
	"unsafe"

	"github.com/peng225/oval/internal/object"
)

// GenerateItem and ValidItem carry the arguments of one Generate / Valid call.
type GenerateItem struct {
	DataSize   int
	WorkerID   int
	BucketName string
	Obj        *object.Object
}

type ValidItem struct {
	WorkerID           int
	ExpectedBucketName string
	Obj                *object.Object
	Body               []byte // what the reader would deliver
}

const batchErrStride = 8192

// cBatch holds the C-side argument arrays of a batch call.  Pointers to Go
// memory stored in them are pinned for the duration of the call.
type cBatch struct {
	pin     runtime.Pinner
	allocs  []unsafe.Pointer
	buckets **C.char
	keys    **C.char
	blens   *C.size_t
	klens   *C.size_t
	wids    *C.int64_t
	wcs     *C.int64_t
	sizes   *C.int64_t
	rcs     *C.int
	errs    *C.char
}

func (b *cBatch) array(n int, elem uintptr) unsafe.Pointer {
	p := C.calloc(C.size_t(n), C.size_t(elem))
	b.allocs = append(b.allocs, p)
	return p
}

func (b *cBatch) free() {
	b.pin.Unpin()
	for _, p := range b.allocs {
		C.free(p)
	}
}

func (b *cBatch) pinned(s string) *C.char {
	if len(s) == 0 {
		return nil
	}
	p := unsafe.StringData(s)
	b.pin.Pin(p)
	return (*C.char)(unsafe.Pointer(p))
}

func newBatch(n int) *cBatch {
	b := &cBatch{}
	ptr := unsafe.Sizeof(uintptr(0))
	b.buckets = (**C.char)(b.array(n, ptr))
	b.keys = (**C.char)(b.array(n, ptr))
	b.blens = (*C.size_t)(b.array(n, unsafe.Sizeof(C.size_t(0))))
	b.klens = (*C.size_t)(b.array(n, unsafe.Sizeof(C.size_t(0))))
	b.wids = (*C.int64_t)(b.array(n, 8))
	b.wcs = (*C.int64_t)(b.array(n, 8))
	b.sizes = (*C.int64_t)(b.array(n, 8))
	b.rcs = (*C.int)(b.array(n, unsafe.Sizeof(C.int(0))))
	b.errs = (*C.char)(b.array(n, batchErrStride))
	return b
}

func (b *cBatch) names(i int, bucket string, obj *object.Object, workerID int) {
	unsafe.Slice(b.buckets, i+1)[i] = b.pinned(bucket)
	unsafe.Slice(b.keys, i+1)[i] = b.pinned(obj.Key)
	unsafe.Slice(b.blens, i+1)[i] = C.size_t(len(bucket))
	unsafe.Slice(b.klens, i+1)[i] = C.size_t(len(obj.Key))
	unsafe.Slice(b.wids, i+1)[i] = C.int64_t(workerID)
	unsafe.Slice(b.wcs, i+1)[i] = C.int64_t(obj.WriteCount)
}

func (b *cBatch) results(n int) []error {
	out := make([]error, n)
	rcs := unsafe.Slice(b.rcs, n)
	for i := 0; i < n; i++ {
		if rcs[i] == 0 {
			continue
		}
		text := C.GoString((*C.char)(unsafe.Add(unsafe.Pointer(b.errs), i*batchErrStride)))
		if rcs[i] < 0 {
			out[i] = fmt.Errorf("oval_hip: %s", text)
		} else {
			out[i] = errors.New(text)
		}
	}
	return out
}

// GenerateBatch is Generate (pattern.go:73-92) for many objects with one device
// launch: data[i], errs[i] are what Generate would have returned for items[i].
func GenerateBatch(items []GenerateItem) ([][]byte, []error) {
	n := len(items)
	if n == 0 {
		return nil, nil
	}
	b := newBatch(n)
	defer b.free()
	outs := (**C.uint8_t)(b.array(n, unsafe.Sizeof(uintptr(0))))
	data := make([][]byte, n)
	for i, it := range items {
		b.names(i, it.BucketName, it.Obj, it.WorkerID)
		unsafe.Slice(b.sizes, n)[i] = C.int64_t(it.DataSize)
		if it.DataSize > 0 {
			data[i] = make([]byte, it.DataSize)
			b.pin.Pin(&data[i][0])
			unsafe.Slice(outs, n)[i] = (*C.uint8_t)(unsafe.Pointer(&data[i][0]))
		} else {
			data[i] = []byte{}
		}
	}
	C.oval_pattern_generate_batch(C.size_t(n), b.sizes, b.wids, b.buckets, b.blens, b.keys, b.klens, b.wcs,
		outs, b.rcs, b.errs, batchErrStride)
	errs := b.results(n)
	for i := range errs {
		if errs[i] != nil {
			data[i] = nil
		}
	}
	return data, errs
}

// ValidBatch is Valid (pattern.go:118-134) for many objects whose bodies are
// already in memory, with one device launch.
func ValidBatch(items []ValidItem) []error {
	n := len(items)
	if n == 0 {
		return nil
	}
	b := newBatch(n)
	defer b.free()
	bodies := (**C.uint8_t)(b.array(n, unsafe.Sizeof(uintptr(0))))
	lens := (*C.size_t)(b.array(n, unsafe.Sizeof(C.size_t(0))))
	for i, it := range items {
		b.names(i, it.ExpectedBucketName, it.Obj, it.WorkerID)
		unsafe.Slice(b.sizes, n)[i] = C.int64_t(it.Obj.Size)
		unsafe.Slice(lens, n)[i] = C.size_t(len(it.Body))
		if len(it.Body) > 0 {
			b.pin.Pin(&it.Body[0])
			unsafe.Slice(bodies, n)[i] = (*C.uint8_t)(unsafe.Pointer(&it.Body[0]))
		}
	}
	C.oval_pattern_valid_batch(C.size_t(n), b.wids, b.buckets, b.blens, b.keys, b.klens, b.sizes, b.wcs,
		bodies, lens, b.rcs, b.errs, batchErrStride)
	return b.results(n)
}
