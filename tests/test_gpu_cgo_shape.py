"""tools/cgo_shape.c on the GPU: a plain-C program, built with gcc against
include/*.h, that calls liboval_hip.so exactly the way the cgo files under
go/internal/pattern do (Go strings as pointer + length, caller-owned output
memory, the io.Reader as a callback, C arrays of pointers for the batch forms,
8192-byte error buffers), from 64 threads at once, and holds every generated
body to the library's generateDataUnit unit by unit.  There is no Go toolchain
in this image; this is the compiled and linked consumer that stands in for it."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_consumer_shaped_like_the_cgo_binding():
    exe = os.path.join(ROOT, "build", "cgo_shape")
    if not os.path.exists(exe):
        # build() makes it; a tree that arrived without build/ gets it here (plain gcc, a second's work)
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-pedantic", "-I" + os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tools", "cgo_shape.c"), "-o", exe, "-L" + os.path.join(ROOT, "oval_amd"),
                               "-loval_hip", "-lpthread", "-Wl,-rpath,$ORIGIN/../oval_amd", "-Wl,-rpath-link,/opt/rocm/lib"])
    p = subprocess.run([exe, "64", "6"], cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = p.stdout.strip().splitlines()[-1]
    assert line.startswith("cgo_shape: 64 threads x 6 rounds") and line.endswith(" 0 failures"), line
    # the reader callback (ovalGoRead in the Go binding) is only ever invoked on the thread that called
    # oval_pattern_valid_stream: cgo can run it as an ordinary Go callback, no foreign thread involved
    assert "(0 on a thread other than the caller's)" in line, line
