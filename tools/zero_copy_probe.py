#!/usr/bin/env python3
"""The two streaming kernels run straight on pinned host memory (the body pointer
handed to oval_fill / oval_validate is a hipHostMalloc buffer, which the GPU can
address): no staging slices, no copy engines, the kernels' own stores and loads
cross the link.  Alone, and a fill and a validate side by side on two contexts
(both directions of the link at once), next to the sliced host-memory calls.

  python tools/zero_copy_probe.py [objects]
"""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oval_amd import _lib, engine  # noqa: E402
from oval_amd.batch import WorkerSpace, span_bytes, synthetic_batch  # noqa: E402

T0 = 1709558129112997


class Side:
    def __init__(self, descs, nbytes):
        L = _lib.lib()
        self.ctx = engine.Context(0)
        self.layout = engine.Layout(self.ctx, descs)
        self.p = C.c_void_p()
        _lib.check(L.oval_pinned_alloc(self.ctx.handle, nbytes, C.byref(self.p)))
        self.host = np.ctypeslib.as_array(C.cast(self.p, C.POINTER(C.c_uint8)), shape=(nbytes,))
        self.res = C.c_void_p()
        self.cnt = C.c_void_p()
        _lib.check(L.oval_dev_alloc(self.ctx.handle, 16 * len(descs), C.byref(self.res)))
        _lib.check(L.oval_dev_alloc(self.ctx.handle, 8 * 16, C.byref(self.cnt)))
        self.nbytes = nbytes

    def fill(self):
        L = _lib.lib()
        _lib.check(L.oval_fill(self.ctx.handle, self.layout._h, self.p, self.nbytes))
        self.ctx.sync()

    def validate(self):
        L = _lib.lib()
        _lib.check(L.oval_validate(self.ctx.handle, self.layout._h, self.p, self.nbytes, self.res, self.cnt))
        self.ctx.sync()


def timed(fn):
    t = time.perf_counter()
    fn()
    return time.perf_counter() - t


def main():
    n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    n_obj -= n_obj % 64
    descs = synthetic_batch(WorkerSpace(n_obj, 64, start_worker_id=0x561B), range(64), "test-bucket",
                            4096, 16384, seed=1, unix_micro=T0)
    nbytes = span_bytes(descs)
    a, b = Side(descs, nbytes), Side(descs, nbytes)
    for s in (a, b):
        s.fill()
        s.validate()
    # what the kernels wrote over the link is what the sliced path writes
    ref = np.empty(nbytes, dtype=np.uint8)
    a.ctx.fill_batch_host(descs, ref)
    assert np.array_equal(ref, a.host), "zero-copy fill differs from the sliced fill"
    tf = float(np.median([timed(a.fill) for _ in range(4)]))
    tv = float(np.median([timed(b.validate) for _ in range(4)]))
    walls = []
    for _ in range(4):
        ths = [threading.Thread(target=a.fill), threading.Thread(target=b.validate)]
        t = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        walls.append(time.perf_counter() - t)
    wall = float(np.median(walls))
    print(f"# kernels straight on pinned host memory, {n_obj} objects 4k-16k, {nbytes / 1e9:.2f} GB per pass")
    print("# fill alone GB/s   validate alone GB/s   fill and validate side by side, both GB/s")
    print("%17.1f %21.1f %42.1f" % (nbytes / tf / 1e9, nbytes / tv / 1e9, 2 * nbytes / wall / 1e9))


if __name__ == "__main__":
    main()
