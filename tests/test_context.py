"""ObjectMeta and ExecutionContext (internal/object/object.go:27-139,
internal/runner/runner.go:26-35,85-135,250-273) — host logic, no GPU.

Pinned by the reference's own output where it has any: the worker key ranges
and worker IDs of README.md:71-74 and README.md:215-218,229-232.  The JSON text
has no fixture in the reference (no saved context is checked in); it is
compared with an independent restatement of encoding/json's rules built on
Python's json module: parity unpinned beyond the struct tags.
"""
import json

import numpy as np
import pytest

from oval_amd import context as X
from oval_amd.batch import build_descs, make_keys

T0 = 1709558129112997


def go_json(obj) -> bytes:
    """encoding/json.Marshal of plain data: compact, UTF-8 kept, HTML-safe."""
    s = json.dumps(obj, separators=(",", ":"), ensure_ascii=False)
    for a, b in (("<", "\\u003c"), (">", "\\u003e"), ("&", "\\u0026"), (" ", "\\u2028"), (" ", "\\u2029")):
        s = s.replace(a, b)
    return s.encode()


def expected_tree(endpoint, buckets, num_obj, num_worker, lo, hi, start, runner_id, state=None):
    """What Runner.init builds (runner.go:106-125), as plain data in struct order."""
    per = num_obj // num_worker
    workers = []
    for w in range(num_worker):
        bwo = []
        for b, name in enumerate(buckets):
            off = (runner_id << 32) + (w << 24)
            objs = [{"key": "ov%010x" % (off + j), "size": 0, "writeCount": 0} for j in range(per)]
            existing = []
            if state and (w, b) in state:
                for j, size, wc in state[(w, b)]:
                    objs[j]["size"], objs[j]["writeCount"] = size, wc
                    existing.append(j)
            bwo.append({"bucketName": name,
                        "objectMeta": {"objectList": objs, "existingObjectIDs": existing, "keyIDOffset": off,
                                       "KeyPrefix": ("ov%010x" % off)[:6]}})
        workers.append({"bucketsWithObject": bwo})
    return {"endpoint": endpoint, "bucketNames": list(buckets), "numObj": num_obj, "numWorker": num_worker,
            "minSize": lo, "maxSize": hi, "startWorkerID": start, "workers": workers}


def test_fresh_context_has_the_readme_key_ranges():
    # README.md:71-74 — --num_obj 1024 --num_worker 4, runner 0, first worker ID 0x614
    ec = X.ExecutionContext.New("http://localhost:9000", ["test-bucket", "test-bucket2"], 1024, 4, 4096, 16384,
                                startWorkerID=0x614, runnerID=0)
    assert [w.id for w in ec.Workers] == [0x614, 0x615, 0x616, 0x617]
    heads_tails = [w.BucketsWithObject[0].ObjectMeta.GetHeadAndTailKey() for w in ec.Workers]
    assert heads_tails == [("ov0000000000", "ov00000000ff"), ("ov0001000000", "ov00010000ff"),
                           ("ov0002000000", "ov00020000ff"), ("ov0003000000", "ov00030000ff")]
    for w in ec.Workers:
        assert [b.BucketName for b in w.BucketsWithObject] == ["test-bucket", "test-bucket2"]
        assert (w.minSize, w.maxSize) == (4096, 16384)
        # the key range is the same for all buckets (worker.go:33-34)
        assert w.BucketsWithObject[1].ObjectMeta.GetHeadAndTailKey() == w.BucketsWithObject[0].ObjectMeta.GetHeadAndTailKey()
        assert len(w.BucketsWithObject[0].ObjectMeta) == 256
    assert ec.Workers[2].BucketsWithObject[0].ObjectMeta.KeyPrefix == "ov0002"
    assert ec.CountExisting() == 0


def test_worker_ids_wrap_and_runner_id_shifts_keys():
    ec = X.ExecutionContext.New("e", ["b"], 6, 3, 256, 512, startWorkerID=0xFFFF, runnerID=0x2A)
    assert [w.id for w in ec.Workers] == [0xFFFF, 0, 1]          # runner.go:112
    om = ec.Workers[1].BucketsWithObject[0].ObjectMeta
    assert om.KeyIDOffset == (0x2A << 32) + (1 << 24)            # runner.go:121-123
    assert om.GetHeadAndTailKey() == ("ov2a01000000", "ov2a01000001")
    # drawn at random when not given (runner.go:109)
    ids = {X.ExecutionContext.New("e", ["b"], 2, 1, 256, 256).StartWorkerID for _ in range(8)}
    assert all(0 <= i < X.maxWorkerID for i in ids) and len(ids) > 1


def test_marshal_is_go_json():
    buckets = ["test-bucket", "b<2>&\"q\"\\"]
    ec = X.ExecutionContext.New("http://localhost:9000/é ", buckets, 4, 2, 4096, 16384,
                                startWorkerID=65535, runnerID=1)
    assert ec.Marshal() == go_json(expected_tree("http://localhost:9000/é ", buckets, 4, 2, 4096, 16384,
                                                 65535, 1))
    # with state: the worker's mutations (worker.go:82-84) and registrations show up
    om = ec.Workers[1].BucketsWithObject[0].ObjectMeta
    o = om.Object(1)
    o.Size = 8192
    o.WriteCount += 3
    om.RegisterToExistingList(o.Key)
    om.RegisterToExistingList(om.Object(0).Key)
    om.Object(0).Size = 4096
    om.Object(0).WriteCount = 1
    tree = expected_tree("http://localhost:9000/é ", buckets, 4, 2, 4096, 16384, 65535, 1,
                         state={(1, 0): [(1, 8192, 3), (0, 4096, 1)]})
    assert ec.Marshal() == go_json(tree)
    # control characters: \n \r \t short forms, the rest \u00XX
    ec2 = X.ExecutionContext.New("a\nb\rc\td\x01e\x7f", ["b"], 1, 1, 256, 256, startWorkerID=0)
    assert b'"endpoint":"a\\nb\\rc\\td\\u0001e\x7f"' in ec2.Marshal()
    # backspace and form feed in the short forms of Go >= 1.22 (go.mod: go 1.24)
    ec3 = X.ExecutionContext.New("a\bb\fc", ["b"], 1, 1, 256, 256, startWorkerID=0)
    assert b'"endpoint":"a\\bb\\fc"' in ec3.Marshal()


def test_load_round_trip_and_save_file(tmp_path):
    X.seed_random(11)
    ec = X.ExecutionContext.New("http://s3:9000", ["bucket-a", "bucket-b"], 64, 4, 4096, 16384, startWorkerID=0x50E4)
    for w in ec.Workers:
        for b in w.BucketsWithObject:
            for _ in range(10):
                o = b.ObjectMeta.GetRandomObject()
                o.Size = 4096 + 256 * (o.index % 5)
                o.WriteCount += 1
                b.ObjectMeta.RegisterToExistingList(o.Key)
    text = ec.Marshal()
    path = tmp_path / "test.json"
    ec.SaveContext(str(path))
    assert path.read_bytes() == text
    for ec2 in (X.ExecutionContext.Load(text), X.ExecutionContext.LoadFile(str(path))):
        assert ec2.Marshal() == text
        assert [w.id for w in ec2.Workers] == [0x50E4, 0x50E5, 0x50E6, 0x50E7]      # README.md:229-232
        assert ec2.CountExisting() == ec.CountExisting() > 0
        # TidyUp rebuilt the membership set (runner.go:126-129)
        for w, w2 in zip(ec.Workers, ec2.Workers):
            for b, b2 in zip(w.BucketsWithObject, w2.BucketsWithObject):
                ids = b.ObjectMeta.ExistingObjectIDs
                assert b2.ObjectMeta.ExistingObjectIDs == ids
                for j in range(len(b.ObjectMeta)):
                    assert b2.ObjectMeta.Exist(b.ObjectMeta.Object(j).Key) == (j in ids)
    # python's json agrees on the meaning of the text
    tree = json.loads(text)
    assert tree["startWorkerID"] == 0x50E4 and len(tree["workers"]) == 4
    assert tree["workers"][3]["bucketsWithObject"][1]["objectMeta"]["KeyPrefix"] == "ov0003"


def test_load_follows_encoding_json_rules():
    text = b"""
    { "EndPoint" : "x\\u00e9\\ud83d\\ude00\\n" , "unknown": {"a":[1,2,{"b":null}], "c":"}"},
      "bucketnames": ["b1", "b\\u003c2"], "NUMOBJ": 2, "numWorker": 1, "minSize": null, "maxSize": 512,
      "startWorkerID": 65535, "extra": 1.5e3,
      "workers": [ { "bucketsWithObject": [ { "BUCKETNAME": "b1", "objectMeta": {
            "existingObjectIDs": [1], "objectList": [ {"key":"ov0000000000","size":0,"writeCount":0},
                                                      {"writeCount":7,"size":512,"key":"ov0000000001","x":[]} ],
            "keyprefix": "ov0000", "keyIDOffset": 0 } } ] } ] }
    """
    ec = X.ExecutionContext.Load(text)
    assert ec.Endpoint == "xé\U0001F600\n"
    assert (ec.NumObj, ec.NumWorker, ec.MinSize, ec.MaxSize, ec.StartWorkerID) == (2, 1, 0, 512, 65535)
    om = ec.Workers[0].BucketsWithObject[0].ObjectMeta
    assert ec.Workers[0].id == 65535 and ec.Workers[0].maxSize == 512
    o = om.Object(1)
    assert (o.Key, o.Size, o.WriteCount) == ("ov0000000001", 512, 7)
    assert om.Exist("ov0000000001") and not om.Exist("ov0000000000")
    assert om.KeyPrefix == "ov0000"
    out = json.loads(ec.Marshal())
    assert out["bucketNames"] == ["b1", "b<2"] and out["endpoint"] == "xé\U0001F600\n"

    for bad, what in [(b'{"bucketNames":["b"],"numObj":1.5}', "non-integer"),
                      (b'{"bucketNames":["b"],"numObj":"3"}', "numObj"),
                      (b'{"bucketNames":["b"]', "unexpected end of JSON input"),
                      (b'{"bucketNames":["b"] "x":1}', "expected"),
                      (b'{"bucketNames":["b"]} x', "after top-level"),
                      (b'{"bucketNames":[]}', "bucket list is empty."),
                      (b'{}', "bucket list is empty."),
                      (b'{"bucketNames":["b"],"workers":[{"bucketsWithObject":[{"objectMeta":{"objectList":[null]}}]}]}', "nil"),
                      (b'{"bucketNames":["b"],"workers":[{"bucketsWithObject":[{"objectMeta":{"objectList":[],"existingObjectIDs":[0]}}]}]}', "outside the object list"),
                      (b'', "unexpected end of JSON input")]:
        with pytest.raises(X.ContextError) as e:
            X.ExecutionContext.Load(bad)
        assert what in str(e.value), (bad, str(e.value))
    with pytest.raises(X.ContextError):
        X.ExecutionContext.LoadFile("/nonexistent/test.json")
    with pytest.raises(X.ContextError):
        X.ExecutionContext.New("e", [], 1, 1, 256, 256)


def test_object_meta_bookkeeping():
    X.seed_random(3)
    om = X.NewObjectMeta(100, (5 << 32) + (2 << 24))
    assert len(om) == 100 and om.KeyPrefix == "ov0502" and om.ExistingObjectIDs == []
    assert om.GetHeadAndTailKey() == ("ov0502000000", "ov0502000063")
    assert om.PopExistingRandomObject() is None and om.GetExistingRandomObject() is None      # object.go:95-97,113-115
    picks = {om.GetRandomObject().index for _ in range(400)}
    assert picks <= set(range(100)) and len(picks) > 80

    for j in (7, 3, 99, 7, 3):                                   # duplicates are ignored (object.go:83-86)
        om.RegisterToExistingList("ov0502%06x" % j)
    assert om.ExistingObjectIDs == [7, 3, 99]
    assert om.Exist("ov0502000063") and om.Exist("ov0502000007") and not om.Exist("ov0502000008")
    # only the six hex digits behind the prefix count (object.go:52-55)
    assert om.Exist("zzzzzz000003") and not om.Exist("ov0502ffffff")
    assert {om.GetExistingRandomObject().index for _ in range(100)} == {3, 7, 99}
    assert len(om.ExistingObjectIDs) == 3

    popped = []
    while True:
        o = om.PopExistingRandomObject()
        if o is None:
            break
        assert not om.Exist(o.Key)
        popped.append(o.index)
    assert sorted(popped) == [3, 7, 99] and om.ExistingObjectIDs == []

    o = om.Object(5)
    o.Size, o.WriteCount = 1024, 9
    assert repr(o) == "&{ov0502000005 1024 9}"                   # how worker.go:99 prints obj
    o.Clear()
    assert (o.Size, o.WriteCount) == (0, 0)

    for bad in ("ov0502", "ov0502xyz", "short", "ov0502-"):
        with pytest.raises(X.ContextError) as e:
            om.Exist(bad)
        assert "strconv.ParseInt" in str(e.value)
    with pytest.raises(X.ContextError):
        om.RegisterToExistingList("ov0502000064")                # ID 100 of a 100-object list
    with pytest.raises(IndexError):
        om.Object(100)


def test_pop_is_uniform_swap_with_last():
    # object.go:102-104: the popped slot is refilled with the last ID
    X.seed_random(1)
    om = X.NewObjectMeta(16, 0)
    for j in range(16):
        om.RegisterToExistingList("ov0000%06x" % j)
    before = om.ExistingObjectIDs
    o = om.PopExistingRandomObject()
    after = om.ExistingObjectIDs
    i = before.index(o.index)
    expect = before[:-1]
    if i < len(expect):
        expect[i] = before[-1]
    assert after == expect


def test_existing_descriptors_match_build_descs():
    X.seed_random(5)
    ec = X.ExecutionContext.New("e", ["test-bucket", "a-bucket-name-longer-than-16"], 40, 4, 256, 4096,
                                startWorkerID=0xFFFE, runnerID=1)
    rng = np.random.default_rng(0)
    for w in ec.Workers:
        for b in w.BucketsWithObject:
            for _ in range(6):
                o = b.ObjectMeta.GetRandomObject()
                o.Size = int(rng.integers(1, 17)) * 256
                o.WriteCount += 1
                b.ObjectMeta.RegisterToExistingList(o.Key)
    descs, refs, nbytes = ec.BuildExistingDescriptors(T0)
    assert len(descs) == len(refs) == ec.CountExisting()
    # the same table from the Python builders, bucket by bucket
    off = 0
    k = 0
    for wi, w in enumerate(ec.Workers):
        for bi, b in enumerate(w.BucketsWithObject):
            ids = b.ObjectMeta.ExistingObjectIDs
            objs = [b.ObjectMeta.Object(j) for j in ids]
            d = build_descs(b.BucketName, [o.Key for o in objs], [o.WriteCount for o in objs], w.id, T0,
                            [o.Size for o in objs])
            d["body_offset"] += np.uint64(off)
            got = descs[k:k + len(ids)]
            assert got.tobytes() == d.tobytes()
            assert (refs["worker"][k:k + len(ids)] == wi).all() and (refs["bucket"][k:k + len(ids)] == bi).all()
            assert refs["obj_id"][k:k + len(ids)].tolist() == ids
            off += int(sum(o.Size for o in objs))
            k += len(ids)
    assert nbytes == off
    # a rank's share (workers [1,3)) is the matching slice, rebased to offset 0
    d13, r13, n13 = ec.BuildExistingDescriptors(T0, 1, 3)
    sel = (refs["worker"] >= 1) & (refs["worker"] < 3)
    assert r13.tobytes() == refs[sel].tobytes()
    base = descs["body_offset"][sel][0]
    assert (d13["body_offset"] == descs["body_offset"][sel] - base).all()
    assert n13 == int((descs["n_units"][sel].astype(np.int64) * 256).sum())


def test_populate_existing_is_seeded_and_decide_size_shaped():
    def make(seed, frac):
        ec = X.ExecutionContext.New("e", ["b1", "b2"], 4000, 4, 4096, 16384, startWorkerID=7)
        ec.PopulateExisting(seed, frac, 1000)
        return ec
    a, b, c = make(9, 0.5), make(9, 0.5), make(10, 0.5)
    assert a.Marshal() == b.Marshal() != c.Marshal()
    n = a.CountExisting()
    assert 0.45 * 8000 < n < 0.55 * 8000
    descs, refs, nbytes = a.BuildExistingDescriptors(T0)
    sizes = descs["n_units"].astype(np.int64) * 256
    assert sizes.min() >= 4096 and sizes.max() <= 16384 + 256      # the reference's own upper quirk included
    assert 1 <= descs["write_count"].min() and descs["write_count"].max() <= 1000
    assert 8500 < sizes.mean() < 10500                             # ~9.5 KB mean for 4k-16k
    full = make(1, 1.0)
    assert full.CountExisting() == 8000
    with pytest.raises(X.ContextError):
        full.PopulateExisting(1, 1.0, 0)


def test_invalid_utf8_becomes_replacement_character():
    # encoding/json: invalid UTF-8 inside a string decodes to U+FFFD, and Marshal writes invalid bytes as �
    ec = X.ExecutionContext.Load(b'{"endpoint":"a\xffb\xc3(\xe2\x82","bucketNames":["b"]}')
    assert ec.Endpoint == "a�b�(��"
    assert b'"endpoint":"a\xef\xbf\xbdb\xef\xbf\xbd(\xef\xbf\xbd\xef\xbf\xbd"' in ec.Marshal()
    ec2 = X.ExecutionContext.New(b"x\xffy", ["b"], 1, 1, 256, 256, startWorkerID=0)
    assert b'"endpoint":"x\\ufffdy"' in ec2.Marshal()


def test_existing_descriptors_at_a_size_that_takes_several_threads():
    """400 K existing objects (the table is then built segment by segment on
    several host threads): offsets chain across segments exactly as in the
    sequential order, every descriptor equals what oval_desc_init gives."""
    X.seed_random(11)
    ec = X.ExecutionContext.New("e", ["test-bucket", "bucket-two"], 400_000, 16, 4096, 16384, startWorkerID=7)
    ec.PopulateExisting(seed=3, fraction=1.0, max_write_count=9)
    descs, refs, nbytes = ec.BuildExistingDescriptors(T0)
    assert len(descs) == ec.CountExisting() == 800_000
    units = descs["n_units"].astype(np.uint64)
    starts = np.concatenate([[0], np.cumsum(units * np.uint64(256))[:-1]]).astype(np.uint64)
    assert (descs["body_offset"] == starts).all() and nbytes == int((units * np.uint64(256)).sum())
    order = np.lexsort((np.arange(len(refs)), refs["bucket"], refs["worker"]))
    assert (order == np.arange(len(refs))).all()                      # worker-major, bucket, then list order
    # spot-check segments against the Python builder
    for wi, bi in ((0, 0), (7, 1), (15, 1)):
        b = ec.Workers[wi].BucketsWithObject[bi]
        sel = np.flatnonzero((refs["worker"] == wi) & (refs["bucket"] == bi))
        ids = refs["obj_id"][sel].tolist()
        assert ids == b.ObjectMeta.ExistingObjectIDs
        objs = [b.ObjectMeta.Object(j) for j in ids[:500]]
        d = build_descs(b.BucketName, [o.Key for o in objs], [o.WriteCount for o in objs], ec.Workers[wi].id, T0,
                        [o.Size for o in objs])
        d["body_offset"] += descs["body_offset"][sel[0]]
        assert descs[sel[:500]].tobytes() == d.tobytes()


def test_object_entries_in_any_shape_load_alike():
    """The reader has a short cut for object entries exactly as encoding/json
    writes them; entries in any other legal shape (white space, another member
    order, escapes in the key, unknown members, 19-digit numbers) take the general
    route and must give the same objects."""
    import json

    def ctx(entries):
        return ('{"endpoint":"e","bucketNames":["b"],"numObj":4,"numWorker":1,"minSize":256,"maxSize":512,'
                '"startWorkerID":3,"workers":[{"bucketsWithObject":[{"bucketName":"b","objectMeta":{"objectList":['
                + ",".join(entries) + '],"existingObjectIDs":[0,2],"keyIDOffset":0,"KeyPrefix":"ov0000"}}]}]}').encode()

    compact = ['{"key":"ov0000000000","size":512,"writeCount":7}', '{"key":"ov0000000001","size":0,"writeCount":0}',
               '{"key":"ov0000000002","size":256,"writeCount":922337203685477580}',
               '{"key":"ov0000000003","size":-256,"writeCount":1}']
    other = ['{ "key" : "ov0000000000" , "size" : 512 , "writeCount" : 7 }',
             '{"writeCount":0,"size":0,"key":"ov000000000\\u0031","extra":[1,{"a":null}]}',
             '{"key":"ov0000000002","size":256,"writeCount":922337203685477580,"KEY":"ov0000000002"}',
             '{"key":"ov0000000003","size":-256,"writeCount":1}\n']
    a = X.ExecutionContext.Load(ctx(compact))
    b = X.ExecutionContext.Load(ctx(other))
    assert a.Marshal() == b.Marshal()
    tree = json.loads(a.Marshal())
    objs = tree["workers"][0]["bucketsWithObject"][0]["objectMeta"]["objectList"]
    assert [o["key"] for o in objs] == ["ov000000000%d" % i for i in range(4)]
    assert [o["size"] for o in objs] == [512, 0, 256, -256] and objs[2]["writeCount"] == 922337203685477580
    # numbers beyond int64 and non-integers are errors on either route
    for bad in ('{"key":"k","size":92233720368547758070,"writeCount":1}', '{"key":"k","size":1.5,"writeCount":1}',
                '{"key":"k","size":1e3,"writeCount":1}'):
        with pytest.raises(Exception):
            X.ExecutionContext.Load(ctx([bad]))


def test_large_context_reads_its_workers_side_by_side_like_in_order():
    """A saved context of several megabytes is read worker by worker on several
    threads when it has the shape encoding/json writes; a space after the opening
    bracket of the workers array sends the same text down the in-order route.
    Both must give the same context, and every text that does not fit the
    side-by-side route (separators with white space, a look-alike object nested in
    an unknown member, a syntax error, a bad ID list) must behave as in order."""
    X.seed_random(5)
    ec = X.ExecutionContext.New("http://e:9000", ["test-bucket", "bucket-two"], 100_000, 8, 4096, 16384, startWorkerID=65530)
    ec.PopulateExisting(seed=9, fraction=0.7, max_write_count=50)
    text = ec.Marshal()
    assert len(text) > (8 << 20)
    in_order = text.replace(b'"workers":[', b'"workers":[ ', 1)
    a, b = X.ExecutionContext.Load(text), X.ExecutionContext.Load(in_order)
    assert a.Marshal() == text and b.Marshal() == text
    assert [w.id for w in a.Workers] == [w.id for w in b.Workers] == [(65530 + i) % 65536 for i in range(8)]
    assert a.CountExisting() == b.CountExisting() == ec.CountExisting()
    da, db = a.BuildExistingDescriptors(T0), b.BuildExistingDescriptors(T0)
    assert da[0].tobytes() == db[0].tobytes() and da[1].tobytes() == db[1].tobytes()

    # separators with white space between the workers
    spaced = text.replace(b'},{"bucketsWithObject":[', b'} , {"bucketsWithObject":[')
    assert X.ExecutionContext.Load(spaced).Marshal() == text
    # an unknown member that holds an object opening like a worker
    k = text.index(b'"bucketsWithObject":[', len(text) // 2)
    nested = text[:k] + b'"zz":[{"bucketsWithObject":[]},{"bucketsWithObject":[null]}],' + text[k:]
    assert X.ExecutionContext.Load(nested).Marshal() == text
    # a syntax error deep inside: the message (with its offset) is the in-order route's
    cut = text.index(b'"size":', 2 * len(text) // 3)
    broken = text[:cut] + b'"size";' + text[cut + 7:]
    msgs = []
    for t in (broken, broken.replace(b'"workers":[', b'"workers":[ ', 1)):
        with pytest.raises(Exception) as ei:
            X.ExecutionContext.Load(t)
        msgs.append(str(ei.value))
    assert f"offset {cut + 6}" in msgs[0] and msgs[1] == msgs[0].replace(f"offset {cut + 6}", f"offset {cut + 7}")
    # IDs outside the object list in two workers: the first worker's is the one reported
    tree = json.loads(text)
    tree["workers"][2]["bucketsWithObject"][1]["objectMeta"]["existingObjectIDs"][5] = 777777
    tree["workers"][6]["bucketsWithObject"][0]["objectMeta"]["existingObjectIDs"][0] = 888888
    bad = json.dumps(tree, separators=(",", ":")).encode()
    for t in (bad, bad.replace(b'"workers":[', b'"workers":[ ', 1)):
        with pytest.raises(Exception, match="777777"):
            X.ExecutionContext.Load(t)


def test_save_file_equals_marshal_for_a_large_context(tmp_path):
    X.seed_random(6)
    ec = X.ExecutionContext.New("e", ["b"], 120_000, 6, 4096, 16384, startWorkerID=1)
    ec.PopulateExisting(seed=2, fraction=1.0, max_write_count=3)
    path = str(tmp_path / "ctx.json")
    ec.SaveContext(path)
    assert open(path, "rb").read() == ec.Marshal()
    assert X.ExecutionContext.LoadFile(path).Marshal() == ec.Marshal()


def _expand_compact(segs, recs):
    """Descriptors from compact records, the way the device writes them."""
    from oval_amd._lib import DESC_DTYPE

    d = np.zeros(len(recs), dtype=DESC_DTYPE)
    units = recs["n_units"].astype(np.uint64)
    d["body_offset"] = (np.concatenate([[0], np.cumsum(units)[:-1]]) * np.uint64(256)).astype(np.uint64) if len(recs) else 0
    d["n_units"] = recs["n_units"]
    d["write_count"] = recs["write_count"]
    d["key_len"] = 12
    for s in segs:
        a, b = int(s["first_obj"]), int(s["first_obj"]) + int(s["n_obj"])
        d["bucket"][a:b] = s["bucket"]
        d["bucket_len"][a:b] = s["bucket_len"]
        d["flags"][a:b] = s["flags"]
        d["worker_id"][a:b] = s["worker_id"]
        d["unix_micro"][a:b] = s["unix_micro"]
        ids = recs["obj_id"][a:b].astype(np.uint64) + np.uint64(s["key_id_base"])
        keys = np.array([list(b"ov%010x" % int(i)) for i in ids], dtype=np.uint8).reshape(-1, 12)
        d["key"][a:b] = keys
    return d


def test_compact_table_describes_the_same_objects_as_the_descriptor_table():
    X.seed_random(8)
    ec = X.ExecutionContext.New("e", ["test-bucket", "a-bucket-name-of-20"], 3000, 5, 256, 8192, startWorkerID=65534, runnerID=3)
    ec.PopulateExisting(seed=4, fraction=0.6, max_write_count=40)
    # one worker's second bucket holds nothing: no segment for it
    om = ec.Workers[2].BucketsWithObject[1].ObjectMeta
    while om.PopExistingRandomObject() is not None:
        pass
    descs, refs, nbytes = ec.BuildExistingDescriptors(T0)
    segs, recs, nbytes2 = ec.BuildExistingCompact(T0)
    assert len(segs) == 9 and len(recs) == len(descs) and nbytes2 == nbytes
    assert _expand_compact(segs, recs).tobytes() == descs.tobytes()
    assert (recs["obj_id"] == refs["obj_id"]).all()
    # a rank's share
    d2, _, nb2 = ec.BuildExistingDescriptors(T0, 1, 3)
    s2, r2, nb3 = ec.BuildExistingCompact(T0, 1, 3)
    assert nb2 == nb3 and _expand_compact(s2, r2).tobytes() == d2.tobytes() and int(s2["first_obj"][0]) == 0
    # a key that does not follow from its ID cannot be described in 12 bytes
    tree = json.loads(ec.Marshal())
    tree["workers"][1]["bucketsWithObject"][0]["objectMeta"]["objectList"][
        tree["workers"][1]["bucketsWithObject"][0]["objectMeta"]["existingObjectIDs"][0]]["key"] = "renamed-key"
    odd = X.ExecutionContext.Load(json.dumps(tree))
    odd.BuildExistingDescriptors(T0)
    with pytest.raises(Exception, match="renamed-key"):
        odd.BuildExistingCompact(T0)


def test_compact_form_of_a_synthetic_batch_expands_to_its_descriptors():
    from oval_amd.batch import WorkerSpace, compact_from_descs, synthetic_batch

    space = WorkerSpace(num_obj=6000, num_worker=12, runner_id=2, start_worker_id=0xFFFA)
    descs = synthetic_batch(space, range(3, 9), "test-bucket", 256, 16384, seed=4, unix_micro=T0)
    segs, recs = compact_from_descs(descs)
    assert len(segs) == 6 and int(segs["n_obj"].sum()) == len(descs) == 3000
    assert _expand_compact(segs, recs).tobytes() == descs.tobytes()
    gapped = descs.copy()
    gapped["body_offset"][5:] += 256
    with pytest.raises(ValueError):
        compact_from_descs(gapped)
