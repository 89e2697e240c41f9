#!/usr/bin/env python3
"""Host-inclusive rate of the pattern path: bodies start (validate) and end
(fill) in host memory, as they do behind an S3 socket.  Uses pinned buffers
from oval_pinned_alloc and the one-shot C-ABI calls, which slice the batch and
overlap PCIe copies with the kernels on a second stream.

  python tools/host_inclusive.py [objects] > profiles/r02_host_inclusive.txt
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oval_amd import _lib, engine  # noqa: E402
from oval_amd.batch import WorkerSpace, span_bytes, synthetic_batch  # noqa: E402

T0 = 1709558129112997


def both_directions(descs, nbytes, units):
    import threading

    L = _lib.lib()
    ctxs = [engine.Context(0), engine.Context(0)]
    bufs, ptrs = [], []
    for c in ctxs:
        p = C.c_void_p()
        _lib.check(L.oval_pinned_alloc(c.handle, nbytes, C.byref(p)))
        ptrs.append(p)
        bufs.append(np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,)))
    for c, b in zip(ctxs, bufs):                      # both buffers hold a valid batch; both contexts are sized
        c.fill_batch_host(descs, b)
        c.validate_batch_host(descs, b)
    alone_f, alone_v, pair = [], [], []
    for rep in range(4):
        t = time.perf_counter()
        ctxs[0].fill_batch_host(descs, bufs[0])
        alone_f.append(time.perf_counter() - t)
        t = time.perf_counter()
        ctxs[1].validate_batch_host(descs, bufs[1])
        alone_v.append(time.perf_counter() - t)
        took = [0.0, 0.0]

        def gen():
            t0 = time.perf_counter()
            ctxs[0].fill_batch_host(descs, bufs[0])
            took[0] = time.perf_counter() - t0

        def val():
            t0 = time.perf_counter()
            _, cnt = ctxs[1].validate_batch_host(descs, bufs[1])
            took[1] = time.perf_counter() - t0
            assert cnt[1] == 0 and cnt[2] == units

        ths = [threading.Thread(target=gen), threading.Thread(target=val)]
        t = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        pair.append((time.perf_counter() - t, took[0], took[1]))
    for c, p in zip(ctxs, ptrs):
        _lib.check(L.oval_pinned_free(c.handle, p))
    wall, tf, tv = (float(np.median([x[i] for x in pair])) for i in range(3))
    return (nbytes / float(np.median(alone_f)) / 1e9, nbytes / float(np.median(alone_v)) / 1e9, nbytes / tf / 1e9,
            nbytes / tv / 1e9, 2 * nbytes / wall / 1e9)


def main():
    n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    n_obj -= n_obj % 64
    descs = synthetic_batch(WorkerSpace(n_obj, 64, start_worker_id=0x561B), range(64), "test-bucket",
                            4096, 16384, seed=1, unix_micro=T0)
    nbytes = span_bytes(descs)
    units = int(descs["n_units"].sum())
    ctx = engine.Context(0)
    L = _lib.lib()
    rows = []
    for kind in ("pinned", "pageable"):
        if kind == "pinned":
            p = C.c_void_p()
            _lib.check(L.oval_pinned_alloc(ctx.handle, nbytes, C.byref(p)))
            host = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))
        else:
            host = np.zeros(nbytes, dtype=np.uint8)
        tfs, tvs = [], []
        for rep in range(5):
            t = time.perf_counter()
            ctx.fill_batch_host(descs, host)
            tfs.append(time.perf_counter() - t)
            t = time.perf_counter()
            res, cnt = ctx.validate_batch_host(descs, host)
            tvs.append(time.perf_counter() - t)
            assert cnt[1] == 0 and cnt[2] == units
        # the first call sizes the context's tables and staging ring; the rest reuse them
        tf, tv = float(np.median(tfs[1:])), float(np.median(tvs[1:]))
        rows.append((kind, nbytes / tfs[0] / 1e9, nbytes / tvs[0] / 1e9, nbytes / tf / 1e9, nbytes / tv / 1e9,
                     2 * nbytes / (tf + tv) / 1e9))
        if kind == "pinned":
            _lib.check(L.oval_pinned_free(ctx.handle, p))
    duplex = both_directions(descs, nbytes, units)
    print(f"# host-inclusive pattern rate, {n_obj} objects 4k-16k, {nbytes / 1e9:.2f} GB per pass, MI355X + PCIe Gen5 x16")
    print("# fill = kernel + D2H, validate = H2D + kernel; descriptor upload, chunk table and result download included")
    print("# host buffer   first call: fill  validate   later calls (median of 4): fill GB/s   validate GB/s   combined GB/s")
    for r in rows:
        print("%-12s %17.1f %9.1f %37.1f %14.1f %14.1f" % r)
    print("# both directions of the link at once: one thread generates batch k + 1 into pinned buffer A while another validates batch k")
    print("# from pinned buffer B (two contexts; what workers in different phases of their put-cycles do to the link between them)")
    print("# fill alone GB/s   validate alone GB/s   side by side: fill GB/s   validate GB/s   both GB/s (bytes generated + validated / wall time)")
    print("%17.1f %21.1f %24.1f %15.1f %11.1f" % duplex)


if __name__ == "__main__":
    main()
