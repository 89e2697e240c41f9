"""The reference's smallest workload (README.md:68: --size 4k-16k --num_obj 1024
--num_worker 4, two buckets) driven through the drop-in API on the GPU: the
PUT / GET / DELETE steps of internal/runner/worker.go:38-232 against an
in-memory object store, then the save / load / re-validate-after-blackout
scenario of README.md:200-236 as ONE device batch built from the loaded context.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T0 = 1709558129112997
NONE = 0xFFFFFFFF


class Store(dict):
    """The S3 side: (bucket, key) -> body bytes."""


def _put(w, store, pattern, oracle, rng):
    """worker.go:38-121"""
    bwo = w.BucketsWithObject[int(rng.integers(len(w.BucketsWithObject)))]
    obj = bwo.ObjectMeta.GetRandomObject()
    before = store.get((bwo.BucketName, obj.Key))
    if before is None:
        assert not bwo.ObjectMeta.Exist(obj.Key), "an object has been lost"
    else:
        assert bwo.ObjectMeta.Exist(obj.Key), "an unexpected object was found"
        pattern.Valid(w.id, bwo.BucketName, obj, before)          # validation before write
    size = pattern.DecideSize(w.minSize, w.maxSize)
    obj.Size = size
    bwo.ObjectMeta.RegisterToExistingList(obj.Key)
    obj.WriteCount += 1
    body = pattern.Generate(size, w.id, bwo.BucketName, obj)
    assert body == oracle.generate(size, w.id, bwo.BucketName, obj.Key, obj.WriteCount, T0)
    store[(bwo.BucketName, obj.Key)] = body
    pattern.Valid(w.id, bwo.BucketName, obj, store[(bwo.BucketName, obj.Key)])   # validation after write


def _get(w, store, pattern, rng):
    """worker.go:123-150"""
    bwo = w.BucketsWithObject[int(rng.integers(len(w.BucketsWithObject)))]
    obj = bwo.ObjectMeta.GetExistingRandomObject()
    if obj is None:
        return
    pattern.Valid(w.id, bwo.BucketName, obj, store[(bwo.BucketName, obj.Key)])


def _delete(w, store, pattern, rng):
    """worker.go:181-232"""
    bwo = w.BucketsWithObject[int(rng.integers(len(w.BucketsWithObject)))]
    obj = bwo.ObjectMeta.PopExistingRandomObject()
    if obj is None:
        return
    pattern.Valid(w.id, bwo.BucketName, obj, store[(bwo.BucketName, obj.Key)])   # validation before delete
    del store[(bwo.BucketName, obj.Key)]
    obj.Clear()


def test_worker_loop_then_blackout_revalidation_in_one_batch(oracle):
    import torch

    from oval_amd import context as X
    from oval_amd import engine, pattern

    pattern.set_clock(T0)
    pattern.seed(42)
    X.seed_random(42)
    rng = np.random.default_rng(42)
    ec = X.ExecutionContext.New("http://localhost:9000", ["test-bucket", "test-bucket2"], 1024, 4, 4096, 16384,
                                startWorkerID=0xFFFE)
    store = Store()
    for step in range(1500):
        w = ec.Workers[step % 4]
        r = rng.random()
        if r < 0.5:
            _put(w, store, pattern, oracle, rng)
        elif r < 0.8:
            _get(w, store, pattern, rng)
        else:
            _delete(w, store, pattern, rng)
    assert len(store) == ec.CountExisting() > 100

    # --save / --load (runner.go:250-273, 85-104)
    ec2 = X.ExecutionContext.Load(ec.Marshal())
    descs, refs, nbytes = ec2.BuildExistingDescriptors(T0)
    assert len(descs) == len(store)

    # the GET side of the re-validation: every stored body at its place in one buffer
    host = np.zeros(nbytes, dtype=np.uint8)
    names = []
    for d, r in zip(descs, refs):
        bwo = ec2.Workers[int(r["worker"])].BucketsWithObject[int(r["bucket"])]
        obj = bwo.ObjectMeta.Object(int(r["obj_id"]))
        body = store[(bwo.BucketName, obj.Key)]
        assert len(body) == obj.Size == int(d["n_units"]) * 256
        off = int(d["body_offset"])
        host[off:off + len(body)] = np.frombuffer(body, dtype=np.uint8)
        names.append((ec2.Workers[int(r["worker"])].id, bwo.BucketName, obj))

    ctx = engine.Context.on_torch_stream(0)
    try:
        layout = engine.Layout(ctx, descs)
        body_dev = torch.from_numpy(host).cuda()
        res, cnt = layout.validate(body_dev)
        ctx.sync()
        c = cnt.cpu().numpy()
        assert c[0] == len(descs) and c[1] == 0 and c[2] == nbytes // 256 and c[9] == 0

        # what a blackout can do: one object comes back as its previous generation,
        # one with a flipped bit in the key field of its third unit
        stale_i, flip_i = 5, len(descs) - 3
        wid, bucket, obj = names[stale_i]
        off = int(descs["body_offset"][stale_i])
        old = oracle.generate(obj.Size, wid, bucket, obj.Key, obj.WriteCount - 1, T0 - 5_000_000)
        host2 = host.copy()
        host2[off:off + obj.Size] = np.frombuffer(old, dtype=np.uint8)
        foff = int(descs["body_offset"][flip_i]) + 2 * 256 + 17
        host2[foff] ^= 0x04
        body_dev.copy_(torch.from_numpy(host2))
        res, cnt = layout.validate(body_dev)
        ctx.sync()
        got = engine.results_to_numpy(res, len(descs))
        exp, ecnt = oracle.validate_batch(descs.view(oracle.DESC_DTYPE), host2)
        for f in exp.dtype.names:
            assert np.array_equal(got[f], exp[f]), f
        assert np.array_equal(cnt.cpu().numpy().astype(np.uint64), ecnt)
        bad = np.flatnonzero(got["first_bad_unit"] != NONE).tolist()
        assert bad == [stale_i, flip_i]
        assert got["first_bad_unit"][stale_i] == 0 and got["first_bad_fields"][stale_i] == 4      # write count
        assert got["first_bad_unit"][flip_i] == 2 and got["first_bad_fields"][flip_i] == 2        # key
        # and the message a worker would log for each is the reference's
        for i in bad:
            wid, bucket, obj = names[i]
            o = int(descs["body_offset"][i])
            data = host2[o:o + obj.Size].tobytes()
            with pytest.raises(pattern.PatternError) as e:
                pattern.Valid(wid, bucket, obj, data)
            assert str(e.value) == oracle.valid(wid, bucket, obj.Key, obj.Size, obj.WriteCount, data)[2]
    finally:
        ctx.close()
