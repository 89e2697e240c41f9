//go:build ovalhip

package pattern

/*
#include <stdint.h>
#include <stdlib.h>
#include "oval_pattern.h"
*/
import "C"

import (
	"fmt"
	"unsafe"

	"github.com/peng225/oval/internal/object"
)

// Parts is the result of Generate for a body that leaves in parts, as
// S3Client.multipartUpload sends it (internal/s3client/s3client.go:164-236:
// body[:partSize] at :185, body = body[partSize:] at :204).  The body is
// generated once into pinned memory owned by the library; Part(i) returns a view
// of that memory, not a copy, and waits only until part i has arrived from the
// device, so part 0 can be on the wire while later parts are still landing.
type Parts struct {
	h *C.oval_parts
	n int
}

// GenerateParts replaces Generate (pattern.go:73-92) for callers that upload in
// parts of partSize bytes.  Errors are Generate's.
func GenerateParts(dataSize, workerID int, bucketName string, obj *object.Object, partSize int) (*Parts, error) {
	b, bl := strArg(bucketName)
	k, kl := strArg(obj.Key)
	text := (*C.char)(C.malloc(errTextCap))
	defer C.free(unsafe.Pointer(text))
	var h *C.oval_parts
	rc := C.oval_pattern_generate_parts(C.int64_t(dataSize), C.int64_t(workerID), b, bl, k, kl,
		C.int64_t(obj.WriteCount), C.int64_t(partSize), &h, text, errTextCap)
	if err := toError(rc, text); err != nil {
		return nil, err
	}
	return &Parts{h: h, n: int(C.oval_parts_count(h))}, nil
}

// Count is the number of UploadPart calls the body takes (s3client.go:176-211).
func (p *Parts) Count() int { return p.n }

// Part returns body[i*partSize : min((i+1)*partSize, dataSize)].  The slice is C
// memory: it stays valid until Release(i) or Close and must not be kept beyond.
func (p *Parts) Part(i int) ([]byte, error) {
	if p.h == nil {
		return nil, fmt.Errorf("oval_hip: parts already closed")
	}
	text := (*C.char)(C.malloc(errTextCap))
	defer C.free(unsafe.Pointer(text))
	var data *C.uint8_t
	var n C.size_t
	rc := C.oval_parts_get(p.h, C.int64_t(i), &data, &n, text, errTextCap)
	if err := toError(rc, text); err != nil {
		return nil, err
	}
	if n == 0 {
		return []byte{}, nil
	}
	return unsafe.Slice((*byte)(unsafe.Pointer(data)), int(n)), nil
}

// Release tells the library that part i has been sent.  After the last part
// the pinned buffer goes back to the library's pool.
func (p *Parts) Release(i int) {
	if p.h != nil {
		C.oval_parts_release(p.h, C.int64_t(i))
	}
}

// Close gives up whatever is still held (the abort path, s3client.go:190-203)
// and frees the handle.  Safe to call twice.
func (p *Parts) Close() {
	if p.h != nil {
		C.oval_parts_close(p.h)
		p.h = nil
	}
}
