"""Mirror of the reference's object bookkeeping and saved execution context
(internal/object/object.go:27-139, internal/runner/runner.go:26-35,85-135,250-273,
internal/runner/worker.go:16-29) over the library's C entry points, plus the
descriptor table of every object a context says exists.

The state lives in the native library; these classes are handles with the
reference's method names.  An `*Object` is an `ObjectRef`: (ObjectMeta, index
into ObjectList) with Key / Size / WriteCount properties, so worker code reads
like worker.go (`obj.Size = size; obj.WriteCount += 1`).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import DESC_DTYPE

_ERR_CAP = 1024
maxWorkerID = 0x10000            # runner.go:22-24

REF_DTYPE = np.dtype([("worker", "<u4"), ("bucket", "<u4"), ("obj_id", "<i8")])
assert REF_DTYPE.itemsize == 16


class ContextError(Exception):
    """A Go `error` (or a fatal log line) of internal/object / internal/runner."""


def _raise(rc: int, err) -> None:
    if rc == 0:
        return
    if rc > 0:
        raise ContextError(err.value.decode("utf-8", "replace"))
    _lib.check(rc)


def _b(s) -> bytes:
    return s.encode("utf-8") if isinstance(s, str) else bytes(s)


def seed_random(seed: int) -> None:
    """Seeds the source of rand.Intn behind the random picks."""
    _lib.lib().oval_object_seed_random(seed)


class ObjectRef:
    """A `*object.Object` inside an ObjectMeta (object.go:21-25)."""
    __slots__ = ("_om", "index")

    def __init__(self, om: "ObjectMeta", index: int):
        self._om, self.index = om, index

    def _get(self):
        key = C.create_string_buffer(13)
        size, wc = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().oval_objmeta_get_object(self._om._h, self.index, key, C.byref(size), C.byref(wc)))
        return key.value.decode(), size.value, wc.value

    @property
    def Key(self) -> str:
        return self._get()[0]

    @property
    def Size(self) -> int:
        return self._get()[1]

    @Size.setter
    def Size(self, v: int) -> None:
        _lib.check(_lib.lib().oval_objmeta_set_object(self._om._h, self.index, v, self._get()[2]))

    @property
    def WriteCount(self) -> int:
        return self._get()[2]

    @WriteCount.setter
    def WriteCount(self, v: int) -> None:
        _lib.check(_lib.lib().oval_objmeta_set_object(self._om._h, self.index, self._get()[1], v))

    def Clear(self) -> None:                                   # object.go:35-38
        _lib.check(_lib.lib().oval_objmeta_set_object(self._om._h, self.index, 0, 0))

    def __repr__(self):
        k, s, w = self._get()
        return f"&{{{k} {s} {w}}}"                              # fmt's %v of a *Object


class ObjectMeta:
    """object.go:27-33; `NewObjectMeta(numObj, keyIDOffset)` is the constructor."""

    def __init__(self, numObj: int = 0, keyIDOffset: int = 0, _borrowed=None, _owner=None):
        if _borrowed is not None:
            self._h, self._own, self._owner = _borrowed, False, _owner
        else:
            self._h = C.c_void_p(_lib.lib().oval_objmeta_new(numObj, keyIDOffset))
            if not self._h:
                raise MemoryError("oval_objmeta_new")
            self._own, self._owner = True, None

    def __del__(self):
        try:
            if self._own and self._h:
                _lib.lib().oval_objmeta_free(self._h)
                self._h = None
        except Exception:
            pass

    def _ref(self, index: int):
        return None if index < 0 else ObjectRef(self, index)

    def GetRandomObject(self):                                  # object.go:72-75
        return self._ref(_lib.lib().oval_objmeta_get_random_object(self._h))

    def RegisterToExistingList(self, key) -> None:              # object.go:77-92
        k = _b(key)
        err = C.create_string_buffer(_ERR_CAP)
        _raise(_lib.lib().oval_objmeta_register_to_existing_list(self._h, k, len(k), err, _ERR_CAP), err)

    def PopExistingRandomObject(self):                          # object.go:94-110
        return self._ref(_lib.lib().oval_objmeta_pop_existing_random_object(self._h))

    def GetExistingRandomObject(self):                          # object.go:112-120
        return self._ref(_lib.lib().oval_objmeta_get_existing_random_object(self._h))

    def Exist(self, key) -> bool:                               # object.go:122-130
        k = _b(key)
        out = C.c_int()
        err = C.create_string_buffer(_ERR_CAP)
        _raise(_lib.lib().oval_objmeta_exist(self._h, k, len(k), C.byref(out), err, _ERR_CAP), err)
        return bool(out.value)

    def GetHeadAndTailKey(self):                                # object.go:132-134
        h, t = C.create_string_buffer(13), C.create_string_buffer(13)
        _lib.lib().oval_objmeta_get_head_and_tail_key(self._h, h, t)
        return h.value.decode(), t.value.decode()

    def TidyUp(self) -> None:                                   # object.go:136-141
        err = C.create_string_buffer(_ERR_CAP)
        _raise(_lib.lib().oval_objmeta_tidy_up(self._h, err, _ERR_CAP), err)

    # fields
    @property
    def KeyIDOffset(self) -> int:
        return _lib.lib().oval_objmeta_key_id_offset(self._h)

    @property
    def KeyPrefix(self) -> str:
        buf = C.create_string_buffer(64)
        n = _lib.lib().oval_objmeta_key_prefix(self._h, buf, 64)
        return buf.raw[:n].decode()

    @property
    def ExistingObjectIDs(self):
        n = _lib.lib().oval_objmeta_num_existing(self._h)
        return [_lib.lib().oval_objmeta_existing_id(self._h, i) for i in range(n)]

    def __len__(self):                                          # len(om.ObjectList)
        return _lib.lib().oval_objmeta_num_objects(self._h)

    def Object(self, index: int) -> ObjectRef:                  # om.ObjectList[index]
        if not (0 <= index < len(self)):
            raise IndexError(index)
        return ObjectRef(self, index)


NewObjectMeta = ObjectMeta


class BucketWithObject:
    """worker.go:26-29"""

    def __init__(self, BucketName: str, ObjectMeta: ObjectMeta):
        self.BucketName, self.ObjectMeta = BucketName, ObjectMeta


class Worker:
    """The serialised part of runner.Worker plus its id (worker.go:16-24)."""

    def __init__(self, id: int, minSize: int, maxSize: int, BucketsWithObject):
        self.id, self.minSize, self.maxSize, self.BucketsWithObject = id, minSize, maxSize, BucketsWithObject


class ExecutionContext:
    """runner.go:26-35.  Build one with `New` (a fresh run, runner.go:106-125),
    `Load` / `LoadFile` (loadSavedContext + the load branch of Runner.init)."""

    def __init__(self, handle):
        self._h = handle
        L = _lib.lib()
        v = [C.c_int64() for _ in range(5)]
        nw, nb = C.c_size_t(), C.c_size_t()
        L.oval_execctx_info(self._h, *[C.byref(x) for x in v], C.byref(nw), C.byref(nb))
        self.NumObj, self.NumWorker, self.MinSize, self.MaxSize, self.StartWorkerID = [x.value for x in v]
        buf = C.create_string_buffer(4096)
        n = L.oval_execctx_endpoint(self._h, buf, 4096)
        self.Endpoint = buf.raw[:n].decode("utf-8", "replace")
        self.Workers = []
        for w in range(nw.value):
            buckets = []
            for b in range(L.oval_execctx_worker_buckets(self._h, w)):
                n = L.oval_execctx_bucket_name(self._h, w, b, buf, 4096)
                om = ObjectMeta(_borrowed=C.c_void_p(L.oval_execctx_objmeta(self._h, w, b)), _owner=self)
                buckets.append(BucketWithObject(buf.raw[:n].decode("utf-8", "replace"), om))
            self.Workers.append(Worker(L.oval_execctx_worker_id(self._h, w), self.MinSize, self.MaxSize, buckets))

    def __del__(self):
        try:
            if self._h:
                _lib.lib().oval_execctx_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @staticmethod
    def _adopt(rc, out, err):
        _raise(rc, err)
        return ExecutionContext(out)

    @classmethod
    def New(cls, endpoint: str, bucketNames, numObj: int, numWorker: int, minSize: int, maxSize: int,
            startWorkerID: int = -1, runnerID: int = 0):
        names = [_b(n) for n in bucketNames]
        arr = (C.c_char_p * max(len(names), 1))(*names)
        out, err = C.c_void_p(), C.create_string_buffer(_ERR_CAP)
        rc = _lib.lib().oval_execctx_new(_b(endpoint), C.cast(arr, C.c_void_p), len(names), numObj, numWorker,
                                         minSize, maxSize, startWorkerID, runnerID, C.byref(out), err, _ERR_CAP)
        return cls._adopt(rc, out, err)

    @classmethod
    def Load(cls, text):
        t = _b(text)
        out, err = C.c_void_p(), C.create_string_buffer(_ERR_CAP)
        return cls._adopt(_lib.lib().oval_execctx_load(t, len(t), C.byref(out), err, _ERR_CAP), out, err)

    @classmethod
    def LoadFile(cls, path: str):
        out, err = C.c_void_p(), C.create_string_buffer(_ERR_CAP)
        return cls._adopt(_lib.lib().oval_execctx_load_file(_b(path), C.byref(out), err, _ERR_CAP), out, err)

    def Marshal(self) -> bytes:
        """json.Marshal(execContext), runner.go:257"""
        n = _lib.lib().oval_execctx_marshal(self._h, None, 0)
        buf = C.create_string_buffer(n + 1)
        _lib.lib().oval_execctx_marshal(self._h, buf, n)
        return buf.raw[:n]

    def SaveContext(self, path: str) -> None:                   # runner.go:250-273
        err = C.create_string_buffer(_ERR_CAP)
        _raise(_lib.lib().oval_execctx_save_file(self._h, _b(path), err, _ERR_CAP), err)

    def CountExisting(self) -> int:
        return _lib.lib().oval_execctx_count_existing(self._h)

    def PopulateExisting(self, seed: int, fraction: float = 1.0, max_write_count: int = 1000) -> None:
        """Every object exists with probability `fraction`, sized by DecideSize."""
        err = C.create_string_buffer(_ERR_CAP)
        _raise(_lib.lib().oval_execctx_populate(self._h, seed, fraction, max_write_count, err, _ERR_CAP), err)

    def BuildExistingDescriptors(self, unix_micro: int, worker_begin: int = 0, worker_end: int = None):
        """(descs, refs, body_bytes): one descriptor per object that exists, for
        workers [worker_begin, worker_end), bodies back to back from offset 0."""
        L = _lib.lib()
        if worker_end is None:
            worker_end = len(self.Workers)
        n, nbytes = C.c_uint64(), C.c_uint64()
        err = C.create_string_buffer(_ERR_CAP)
        _raise(L.oval_execctx_build_descs(self._h, unix_micro, worker_begin, worker_end, None, None, 0,
                                          C.byref(n), C.byref(nbytes), err, _ERR_CAP), err)
        descs = np.zeros(n.value, dtype=DESC_DTYPE)
        refs = np.zeros(n.value, dtype=REF_DTYPE)
        if n.value:
            _raise(L.oval_execctx_build_descs(self._h, unix_micro, worker_begin, worker_end, descs.ctypes.data,
                                              refs.ctypes.data, n.value, C.byref(n), C.byref(nbytes), err,
                                              _ERR_CAP), err)
        return descs, refs, nbytes.value

    def BuildExistingCompact(self, unix_micro: int, worker_begin: int = 0, worker_end: int = None):
        """(segs, recs, body_bytes): the same objects, in the same order, as 12-byte
        records plus one segment per worker and bucket, for Layout.reset_compact."""
        from ._lib import REC_DTYPE, SEGMENT_DTYPE

        L = _lib.lib()
        if worker_end is None:
            worker_end = len(self.Workers)
        ns, n, nbytes = C.c_uint64(), C.c_uint64(), C.c_uint64()
        err = C.create_string_buffer(_ERR_CAP)
        _raise(L.oval_execctx_build_compact(self._h, unix_micro, worker_begin, worker_end, None, 0, None, 0,
                                            C.byref(ns), C.byref(n), C.byref(nbytes), err, _ERR_CAP), err)
        segs = np.zeros(ns.value, dtype=SEGMENT_DTYPE)
        recs = np.zeros(n.value, dtype=REC_DTYPE)
        if n.value:
            _raise(L.oval_execctx_build_compact(self._h, unix_micro, worker_begin, worker_end, segs.ctypes.data,
                                                len(segs), recs.ctypes.data, len(recs), C.byref(ns), C.byref(n),
                                                C.byref(nbytes), err, _ERR_CAP), err)
        return segs, recs, nbytes.value
