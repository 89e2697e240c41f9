//go:build ovalhip

// Package pattern, MI355X build: the same exported surface as pattern.go
// (DecideSize, Generate, Valid and the unexported generateDataUnit /
// validDataUnit that pattern_test.go calls), with every byte written and every
// unit checked by the HIP kernels behind liboval_hip.so.
//
// Drop this file next to internal/pattern/pattern.go, add the line
//
//	//go:build !ovalhip
//
// at the top of pattern.go, and build with
//
//	CGO_ENABLED=1 CGO_CFLAGS=-I<repo>/include \
//	CGO_LDFLAGS="-L<repo>/oval_amd -loval_hip -Wl,-rpath,<repo>/oval_amd" \
//	go build -tags ovalhip
//
// Calls from the worker goroutines (internal/runner/worker.go:64,77,85,109,140,200)
// are folded into shared device launches by the library's micro-batcher, which
// init() switches on; nothing in the callers changes.
package pattern

/*
#cgo LDFLAGS: -loval_hip
#include <stdint.h>
#include <stdlib.h>
#include "oval_pattern.h"

// implemented in Go below (//export)
size_t ovalGoRead(void* user, uint8_t* p, size_t n);
*/
import "C"

import (
	"errors"
	"fmt"
	"io"
	"os"
	"runtime/cgo"
	"strconv"
	"unsafe"

	"github.com/peng225/oval/internal/object"
)

const (
	dataUnitSize                          = 256
	dataUnitHeaderSizeWithoutBucketAndKey = 20

	// room for the longest message of validDataUnit: five lines, sixteen dump
	// lines and six annotation lines
	errTextCap = 8192
)

func init() {
	device := 0
	if s := os.Getenv("OVAL_HIP_DEVICE"); s != "" {
		if v, err := strconv.Atoi(s); err == nil {
			device = v
		}
	}
	if rc := C.oval_pattern_init(C.int(device)); rc != 0 {
		panic(fmt.Sprintf("oval_hip: cannot initialise device %d (status %d)", device, int(rc)))
	}
	// Generate / Valid are called one object at a time from up to 255
	// goroutines; let concurrent calls share launches.
	C.oval_pattern_set_batching(1, 0, 0)
}

// toError turns the library's return convention into a Go error: 0 = nil,
// 1 = an error whose text is exactly the reference's, < 0 = the engine failed.
func toError(rc C.int, text *C.char) error {
	if rc == 0 {
		return nil
	}
	if rc < 0 {
		return fmt.Errorf("oval_hip: %s", C.GoString(text))
	}
	return errors.New(C.GoString(text))
}

// strArg passes a Go string as pointer + length without copying.  The library
// does not keep the pointer after the call returns.
func strArg(s string) (*C.char, C.size_t) {
	if len(s) == 0 {
		return nil, 0
	}
	return (*C.char)(unsafe.Pointer(unsafe.StringData(s))), C.size_t(len(s))
}

// DecideSize replaces pattern.go:22-71.
func DecideSize(minSize, maxSize int) (int, error) {
	var size C.int64_t
	text := (*C.char)(C.malloc(256))
	defer C.free(unsafe.Pointer(text))
	rc := C.oval_pattern_decide_size(C.int64_t(minSize), C.int64_t(maxSize), &size, text, 256)
	if err := toError(rc, text); err != nil {
		return 0, err
	}
	return int(size), nil
}

// Generate replaces pattern.go:73-92.  The slice is caller-owned Go memory, as
// PutObject expects (internal/s3client/s3client.go:145-162).
func Generate(dataSize, workerID int, bucketName string, obj *object.Object) ([]byte, error) {
	if dataSize < 0 {
		return nil, fmt.Errorf("negative data size")
	}
	if twoStep && dataSize <= twoStepMaxBody {
		return generateTwoStep(dataSize, workerID, bucketName, obj) // twostep_hip.go: no thread blocked in the library
	}
	data := make([]byte, dataSize)
	var out *C.uint8_t
	if dataSize > 0 {
		out = (*C.uint8_t)(unsafe.Pointer(&data[0]))
	}
	b, bl := strArg(bucketName)
	k, kl := strArg(obj.Key)
	text := (*C.char)(C.malloc(errTextCap))
	defer C.free(unsafe.Pointer(text))
	rc := C.oval_pattern_generate(C.int64_t(dataSize), C.int64_t(workerID), b, bl, k, kl,
		C.int64_t(obj.WriteCount), out, text, errTextCap)
	if err := toError(rc, text); err != nil {
		return nil, err
	}
	return data, nil
}

// generateDataUnit replaces pattern.go:94-116.
func generateDataUnit(unitCount, workerID int, bucketName string, obj *object.Object) ([]byte, error) {
	data := make([]byte, dataUnitSize)
	b, bl := strArg(bucketName)
	k, kl := strArg(obj.Key)
	text := (*C.char)(C.malloc(errTextCap))
	defer C.free(unsafe.Pointer(text))
	rc := C.oval_pattern_generate_data_unit(C.int64_t(unitCount), C.int64_t(workerID), b, bl, k, kl,
		C.int64_t(obj.WriteCount), (*C.uint8_t)(unsafe.Pointer(&data[0])), text, errTextCap)
	if err := toError(rc, text); err != nil {
		return nil, err
	}
	return data, nil
}

// readerBox carries an io.Reader through the C call as a cgo.Handle stored in
// C memory (a Go pointer may not be handed to C and back as void*).
type readerBox struct {
	r io.Reader
}

//export ovalGoRead
func ovalGoRead(user unsafe.Pointer, p *C.uint8_t, n C.size_t) C.size_t {
	h := *(*cgo.Handle)(user)
	box := h.Value().(*readerBox)
	buf := unsafe.Slice((*byte)(unsafe.Pointer(p)), int(n))
	for {
		got, err := box.r.Read(buf)
		if got > 0 {
			return C.size_t(got)
		}
		if err != nil {
			return 0 // io.EOF or a broken body: Valid reports the short read (pattern.go:125-127)
		}
	}
}

// Valid replaces pattern.go:118-134.  The body is read straight into pinned
// staging memory by the library (through ovalGoRead) and validated on the
// device; the returned error text is the reference's.
func Valid(workerID int, expectedBucketName string, obj *object.Object, reader io.Reader) error {
	if twoStep && obj.Size <= twoStepMaxBody {
		return validTwoStep(workerID, expectedBucketName, obj, reader) // twostep_hip.go
	}
	h := cgo.NewHandle(&readerBox{r: reader})
	defer h.Delete()
	user := (*cgo.Handle)(C.malloc(C.size_t(unsafe.Sizeof(h))))
	defer C.free(unsafe.Pointer(user))
	*user = h
	b, bl := strArg(expectedBucketName)
	k, kl := strArg(obj.Key)
	text := (*C.char)(C.malloc(errTextCap))
	defer C.free(unsafe.Pointer(text))
	rc := C.oval_pattern_valid_stream(C.int64_t(workerID), b, bl, k, kl, C.int64_t(obj.Size),
		C.int64_t(obj.WriteCount), C.oval_read_fn(C.ovalGoRead), unsafe.Pointer(user), text, errTextCap)
	return toError(rc, text)
}

// validDataUnit replaces pattern.go:136-180.
func validDataUnit(unitCount, workerID int, expectedBucketName string, obj *object.Object, data []byte) error {
	if len(data) < dataUnitSize {
		return fmt.Errorf("a data unit is %d bytes, got %d", dataUnitSize, len(data))
	}
	b, bl := strArg(expectedBucketName)
	k, kl := strArg(obj.Key)
	text := (*C.char)(C.malloc(errTextCap))
	defer C.free(unsafe.Pointer(text))
	rc := C.oval_pattern_valid_data_unit(C.int64_t(unitCount), C.int64_t(workerID), b, bl, k, kl,
		C.int64_t(obj.WriteCount), (*C.uint8_t)(unsafe.Pointer(&data[0])), text, errTextCap)
	return toError(rc, text)
}

// dump replaces pattern.go:182-206 (kept for callers inside the package).
func dump(data string) string {
	if len(data) == 0 {
		return ""
	}
	capacity := C.size_t(len(data) + 1024)
	out := (*C.char)(C.malloc(capacity))
	defer C.free(unsafe.Pointer(out))
	d, dl := strArg(data)
	n := C.oval_pattern_dump(d, dl, out, capacity)
	return C.GoStringN(out, C.int(n))
}

// SetBatching switches the folding of concurrent calls into shared launches
// (on by default) and sets how long a launch waits for company (0 = not at all).
func SetBatching(enable bool, windowMicros int) {
	e := C.int(0)
	if enable {
		e = 1
	}
	C.oval_pattern_set_batching(e, C.int(windowMicros), 0)
}

// Shutdown releases the device; call it when the runner is done.
func Shutdown() {
	C.oval_pattern_shutdown()
}
