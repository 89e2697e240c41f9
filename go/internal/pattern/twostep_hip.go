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
	"io"
	"os"
	"runtime"
	"unsafe"

	"github.com/peng225/oval/internal/object"
)

// Generate and Valid in two steps.  A blocking cgo call keeps its OS thread for
// as long as the batch is on the device, and with 255 worker goroutines
// (internal/runner/runner.go:186-214) on a 16-core host the threads' sleeps and
// wake-ups cost more than the device work.  Here the goroutine hands the job to
// the library (oval_pattern_generate_begin / oval_pattern_valid_begin return at
// once), yields to the Go scheduler between polls and collects the outcome with
// oval_pattern_poll: no thread ever sleeps inside the library, and the jobs of all
// goroutines that are under way share device launches.  Measured with the same
// call sequence from C (tools/compat_bench.cpp, COMPAT_ASYNC): 1.8 M put-cycles/s
// with 224 workers on 14 threads against 0.84 M with 255 blocking threads.
//
// OVAL_HIP_TWO_STEP=0 in the environment keeps the blocking calls.
var twoStep = os.Getenv("OVAL_HIP_TWO_STEP") != "0"

// twoStepMaxBody is the largest body that can share a launch (the library
// carries out bigger ones inside *_begin, which would block the thread after all).
const twoStepMaxBody = 256 << 10

// await polls h until the call is over; the goroutine yields in between, so its
// thread runs the other workers meanwhile.
func await(h *C.oval_pending, text *C.char) C.int {
	for {
		rc := C.oval_pattern_poll(h, text, errTextCap)
		if rc != C.OVAL_PATTERN_PENDING {
			return rc
		}
		runtime.Gosched()
	}
}

// generateTwoStep is Generate (pattern.go:73-92) without a blocked thread.  The
// output slice is pinned for as long as the library may write into it.
func generateTwoStep(dataSize, workerID int, bucketName string, obj *object.Object) ([]byte, error) {
	data := make([]byte, dataSize)
	var pin runtime.Pinner
	defer pin.Unpin()
	var out *C.uint8_t
	if dataSize > 0 {
		pin.Pin(&data[0])
		out = (*C.uint8_t)(unsafe.Pointer(&data[0]))
	}
	b, bl := strArg(bucketName)
	k, kl := strArg(obj.Key)
	text := (*C.char)(C.malloc(errTextCap))
	defer C.free(unsafe.Pointer(text))
	var h *C.oval_pending
	if rc := C.oval_pattern_generate_begin(C.int64_t(dataSize), C.int64_t(workerID), b, bl, k, kl,
		C.int64_t(obj.WriteCount), out, &h); rc != 0 {
		return nil, fmt.Errorf("oval_hip: generate_begin failed (status %d)", int(rc))
	}
	if err := toError(await(h, text), text); err != nil {
		return nil, err
	}
	return data, nil
}

// validTwoStep is Valid (pattern.go:118-134) for a body of at most
// twoStepMaxBody bytes: the reader is drained into a Go slice first (io.ReadFull
// semantics: a short body is reported as the reference reports it), then the
// complete units are checked on the device while the goroutine yields.
func validTwoStep(workerID int, expectedBucketName string, obj *object.Object, reader io.Reader) error {
	want := obj.Size / dataUnitSize * dataUnitSize
	body := make([]byte, want)
	got, _ := io.ReadFull(reader, body)
	var pin runtime.Pinner
	defer pin.Unpin()
	var p *C.uint8_t
	if got > 0 {
		pin.Pin(&body[0])
		p = (*C.uint8_t)(unsafe.Pointer(&body[0]))
	}
	b, bl := strArg(expectedBucketName)
	k, kl := strArg(obj.Key)
	text := (*C.char)(C.malloc(errTextCap))
	defer C.free(unsafe.Pointer(text))
	var h *C.oval_pending
	if rc := C.oval_pattern_valid_begin(C.int64_t(workerID), b, bl, k, kl, C.int64_t(obj.Size),
		C.int64_t(obj.WriteCount), p, C.size_t(got), &h); rc != 0 {
		return fmt.Errorf("oval_hip: valid_begin failed (status %d)", int(rc))
	}
	return toError(await(h, text), text)
}
