//go:build ovalhip

package pattern

// Tests of what the MI355X build adds to the package (GenerateBatch, ValidBatch,
// GenerateParts) and of Generate / Valid at sizes the reference's own tests do
// not reach.  The reference's pattern_test.go runs unchanged next to this file.
// Needs a gfx950 device:
//
//	CGO_ENABLED=1 CGO_CFLAGS=-I<repo>/include CGO_LDFLAGS="-L<repo>/oval_amd -Wl,-rpath,<repo>/oval_amd" \
//	go test -tags ovalhip ./internal/pattern/

import (
	"bytes"
	"io"
	"strings"
	"sync"
	"testing"

	"github.com/peng225/oval/internal/object"
)

// sameButTime reports whether two bodies agree everywhere except in the unix
// time of each unit (bytes 40..47, which validDataUnit skips, pattern.go:172).
func sameButTime(a, b []byte) bool {
	if len(a) != len(b) {
		return false
	}
	for off := 0; off < len(a); off += dataUnitSize {
		end := off + dataUnitSize
		if end > len(a) {
			end = len(a)
		}
		if !bytes.Equal(a[off:min(off+40, end)], b[off:min(off+40, end)]) {
			return false
		}
		if off+48 < end && !bytes.Equal(a[off+48:end], b[off+48:end]) {
			return false
		}
	}
	return true
}

func testObjects(n int) []*object.Object {
	objs := make([]*object.Object, n)
	for i := range objs {
		objs[i] = object.NewObject(int64(3<<24 + i))
		objs[i].Size = (1 + i%64) * dataUnitSize
		objs[i].WriteCount = 1 + i%9
	}
	return objs
}

func TestGenerateBatchMatchesGenerate(t *testing.T) {
	objs := testObjects(300)
	items := make([]GenerateItem, len(objs))
	for i, o := range objs {
		items[i] = GenerateItem{DataSize: o.Size, WorkerID: 0x561b + i%4, BucketName: "test-bucket", Obj: o}
	}
	data, errs := GenerateBatch(items)
	if len(data) != len(items) || len(errs) != len(items) {
		t.Fatalf("GenerateBatch returned %d bodies and %d errors for %d items", len(data), len(errs), len(items))
	}
	for i, it := range items {
		if errs[i] != nil {
			t.Fatalf("item %d: %v", i, errs[i])
		}
		one, err := Generate(it.DataSize, it.WorkerID, it.BucketName, it.Obj)
		if err != nil {
			t.Fatalf("Generate %d: %v", i, err)
		}
		if !sameButTime(data[i], one) {
			t.Fatalf("item %d: GenerateBatch and Generate differ outside the time field", i)
		}
		if err := Valid(it.WorkerID, it.BucketName, it.Obj, bytes.NewReader(data[i])); err != nil {
			t.Fatalf("Valid of item %d: %v", i, err)
		}
	}
}

func TestValidBatchFindsWhatValidFinds(t *testing.T) {
	objs := testObjects(200)
	items := make([]ValidItem, len(objs))
	for i, o := range objs {
		body, err := Generate(o.Size, 7, "test-bucket", o)
		if err != nil {
			t.Fatal(err)
		}
		switch i % 5 {
		case 1:
			body[(o.Size/dataUnitSize-1)*dataUnitSize+28] ^= 1 // write count of the last unit
		case 2:
			body = body[:len(body)-100] // short read inside the last unit
		}
		items[i] = ValidItem{WorkerID: 7, ExpectedBucketName: "test-bucket", Obj: o, Body: body}
	}
	errs := ValidBatch(items)
	for i, it := range items {
		want := Valid(it.WorkerID, it.ExpectedBucketName, it.Obj, bytes.NewReader(it.Body))
		switch {
		case want == nil && errs[i] != nil:
			t.Fatalf("item %d: ValidBatch reports %v, Valid nothing", i, errs[i])
		case want != nil && errs[i] == nil:
			t.Fatalf("item %d: Valid reports %v, ValidBatch nothing", i, want)
		case want != nil && want.Error() != errs[i].Error():
			t.Fatalf("item %d: messages differ:\n%v\n%v", i, want, errs[i])
		}
		if i%5 == 1 && (errs[i] == nil || !strings.Contains(errs[i].Error(), "WriteCount is wrong")) {
			t.Fatalf("item %d: damaged write count not reported: %v", i, errs[i])
		}
		if i%5 == 2 && (errs[i] == nil || !strings.Contains(errs[i].Error(), "could not read some data")) {
			t.Fatalf("item %d: short read not reported: %v", i, errs[i])
		}
	}
}

func TestGeneratePartsIsGenerateInParts(t *testing.T) {
	obj := object.NewObject(0x20000e2)
	obj.Size = 20<<20 + 3*dataUnitSize
	obj.WriteCount = 2
	const partSize = 5 << 20
	parts, err := GenerateParts(obj.Size, 0x561b, "test-bucket", obj, partSize)
	if err != nil {
		t.Fatal(err)
	}
	defer parts.Close()
	if parts.Count() != 5 {
		t.Fatalf("%d parts, want 5", parts.Count())
	}
	whole := make([]byte, 0, obj.Size)
	for i := 0; i < parts.Count(); i++ {
		p, err := parts.Part(i)
		if err != nil {
			t.Fatal(err)
		}
		whole = append(whole, p...)
		parts.Release(i)
	}
	if len(whole) != obj.Size {
		t.Fatalf("parts add up to %d bytes, want %d", len(whole), obj.Size)
	}
	if err := Valid(0x561b, "test-bucket", obj, bytes.NewReader(whole)); err != nil {
		t.Fatal(err)
	}
	// a reader that delivers a few bytes at a time, damage in the last unit
	whole[len(whole)-dataUnitSize+17] ^= 0x20
	err = Valid(0x561b, "test-bucket", obj, &dribble{data: whole, step: 100003})
	if err == nil || !strings.Contains(err.Error(), "Key name is wrong") {
		t.Fatalf("damaged key in the last unit not reported: %v", err)
	}
}

type dribble struct {
	data []byte
	step int
}

func (d *dribble) Read(p []byte) (int, error) {
	if len(d.data) == 0 {
		return 0, io.EOF
	}
	n := min(len(p), d.step, len(d.data))
	copy(p, d.data[:n])
	d.data = d.data[n:]
	return n, nil
}

func TestConcurrentWorkersShareLaunches(t *testing.T) {
	var wg sync.WaitGroup
	failed := make(chan error, 64)
	for w := 0; w < 64; w++ {
		wg.Add(1)
		go func(w int) {
			defer wg.Done()
			obj := object.NewObject(int64(w<<24 + 1))
			for i := 0; i < 200; i++ {
				size, err := DecideSize(4096, 16384)
				if err != nil {
					failed <- err
					return
				}
				obj.Size = size
				obj.WriteCount++
				body, err := Generate(size, 0x561b+w, "test-bucket", obj)
				if err == nil {
					err = Valid(0x561b+w, "test-bucket", obj, bytes.NewReader(body))
				}
				if err != nil {
					failed <- err
					return
				}
			}
		}(w)
	}
	wg.Wait()
	close(failed)
	for err := range failed {
		t.Fatal(err)
	}
}
