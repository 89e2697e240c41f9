"""Pins the CPU oracle to the reference's own known-answer data
(README.md:97-122, internal/pattern/pattern_test.go:30-255) before anything
else is allowed to trust it."""
import numpy as np
import pytest

T0 = 1709558129112997


def test_readme_unit_bytes(oracle, golden):
    fx = golden["readme"]
    w = fx["written"]
    unit = oracle.generate_data_unit(w["unit_count"], w["worker_id"], w["bucket"], w["key"],
                                     w["write_count"], w["unix_micro"])
    assert unit.hex() == fx["unit_hex"]
    assert w["unix_micro"] == T0


def test_readme_error_message(oracle, golden):
    fx = golden["readme"]
    e = fx["expected_state"]
    mask, text = oracle.valid_data_unit(e["unit_count"], e["worker_id"], e["bucket"], e["key"],
                                        e["write_count"], bytes.fromhex(fx["unit_hex"]))
    assert mask == 4 | 8          # write count and offset, in that order (README.md:98-99)
    assert text == fx["message"]


def test_readme_unit_valid_against_its_own_state(oracle, golden):
    fx = golden["readme"]
    w = fx["written"]
    mask, text = oracle.valid_data_unit(w["unit_count"], w["worker_id"], w["bucket"], w["key"],
                                        w["write_count"], bytes.fromhex(fx["unit_hex"]))
    assert mask == 0 and text is None


def _check_header(unit, pt, bucket_hex, hdr):
    assert len(unit) == 256
    assert unit[0:16].hex() == bucket_hex
    assert unit[16:28].hex() == pt["key_field_hex"]
    assert unit[28:32].hex() == hdr["write_count_bytes"]
    assert unit[32:36].hex() == hdr["offset_bytes"]
    assert unit[36:40].hex() == hdr["worker_bytes"]


def test_generate_data_unit_success(oracle, golden):          # pattern_test.go:30-57
    pt = golden["pattern_test"]
    g = pt["generate_data_unit"]
    unit = oracle.generate_data_unit(g["unit_count"], pt["worker_id"], pt["bucket"], pt["obj"]["key"],
                                     pt["obj"]["write_count"], T0)
    assert len(unit) == g["len"]
    _check_header(unit, pt, pt["bucket_field_hex"], g)


@pytest.mark.parametrize("which", ["generate_512", "generate_long_bucket_512"])
def test_generate_success(oracle, golden, which):             # pattern_test.go:59-155
    pt = golden["pattern_test"]
    long = which.startswith("generate_long")
    bucket = pt["long_bucket"] if long else pt["bucket"]
    data = oracle.generate(512, pt["worker_id"], bucket, pt["obj"]["key"], pt["obj"]["write_count"], T0)
    assert len(data) == 512
    for i, hdr in enumerate(pt[which]):
        _check_header(data[256 * i:256 * (i + 1)], pt,
                      pt["long_bucket_field_hex"] if long else pt["bucket_field_hex"], hdr)


def test_valid_round_trips(oracle, golden):                   # pattern_test.go:157-200
    pt = golden["pattern_test"]
    key, wc, wid = pt["obj"]["key"], pt["obj"]["write_count"], pt["worker_id"]
    for case in pt["valid_round_trips"]:
        if case["what"] == "unit":
            u = oracle.generate_data_unit(case["unit_count"], wid, case["bucket"], key, wc, T0)
            assert oracle.valid_data_unit(case["unit_count"], wid, case["bucket"], key, wc, u) == (0, None)
        else:
            data = oracle.generate(case["size"], wid, case["bucket"], key, wc, T0)
            assert oracle.valid(wid, case["bucket"], key, case["size"], wc, data) == (0, -1, None)


def test_decide_size_domain(oracle, golden):                  # pattern_test.go:202-255
    rng = np.random.default_rng(1)
    for tc in golden["pattern_test"]["decide_size"]:
        if tc["err"]:
            with pytest.raises(oracle.OracleError):
                oracle.decide_size(tc["min"], tc["max"], 0.3)
            continue
        for y in list(rng.random(200)) + [0.0, 1.0 - 2.0 ** -52]:
            s = oracle.decide_size(tc["min"], tc["max"], float(y))
            assert tc["min"] <= s <= tc["max"] and s % 256 == 0


def test_decide_size_top_sample_quirk(oracle):
    # rand.Float64() can return 1 - 2^-53; 1 - 0.5*y then rounds to exactly 0.5,
    # x becomes 1.0 and the formula at pattern.go:70 yields maxSize + 256.  The
    # restatement keeps the arithmetic as written rather than clamping.
    assert oracle.decide_size(512, 512, 1.0 - 2.0 ** -53) == 768


def test_decide_size_error_texts(oracle):                     # pattern.go:23-34
    cases = [((0, 512), "minSize should be larger than or equal to 256"),
             ((513, 1024), "minSize should be a multiple of 256"),
             ((512, 513), "maxSize should be a multiple of 256"),
             ((1024, 512), "maxSize should be larger than or equal to minSize")]
    for (a, b), text in cases:
        with pytest.raises(oracle.OracleError, match=text):
            oracle.decide_size(a, b, 0.5)


def test_decide_size_mean_matches_survey(oracle):
    # mean object size for 4k-16k is about 9.5 KB under pattern.go:63-70
    ys = (np.arange(20000) + 0.5) / 20000
    mean = np.mean([oracle.decide_size(4096, 16384, float(y)) for y in ys])
    assert 9300 < mean < 9700


def test_decide_size_against_an_independent_log2(oracle):
    """The reference holds no vector for the value DecideSize returns for a given
    sample (parity unpinned: math.Log2 is restated from the Go standard library's
    portable implementation, as it rounds on amd64 where Go does not fuse
    multiply-adds).  What can be checked without Go: the restated logarithm agrees
    with an independent one (numpy's) so closely that over 150,000 samples and a
    range of 4 million sizes not one truncated product differs — a wrong
    coefficient or a dropped term of relative size 1e-11 would already show."""
    rng = np.random.default_rng(12)
    ys = rng.random(150_000)
    lo, hi = 256, 1 << 30
    n = (hi - lo) // 256 + 1
    want = lo + 256 * np.floor(np.float64(n) * -np.log2(1.0 - 0.5 * ys)).astype(np.int64)
    got = np.array([oracle.decide_size(lo, hi, float(y)) for y in ys], dtype=np.int64)
    assert int((got != want).sum()) == 0


def test_key_format(oracle):                                  # object.go:48-50, README.md:71-74
    assert oracle.generate_key(0) == "ov0000000000"
    assert oracle.generate_key((2 << 24) + 0xE2) == "ov00020000e2"
    assert oracle.generate_key((1 << 32) + (3 << 24) + 0xFF) == "ov01030000ff"


def test_trim_space_go_semantics(oracle):
    ts = oracle.trim_space
    assert ts(b"  abc \t\r\n") == b"abc"
    assert ts(b"\x0b\x0cabc") == b"abc"
    assert ts(b"\xc2\xa0abc\xc2\x85") == b"abc"                       # U+00A0, U+0085
    assert ts("   abc    　".encode()) == b"abc"
    assert ts(b"\xa0abc") == b"\xa0abc"                               # a bare A0 byte is invalid UTF-8, not a space
    assert ts(b"abc\xc2") == b"abc\xc2"                               # truncated sequence
    assert ts(b"abc\xe2\x80") == b"abc\xe2\x80"
    assert ts(b"\xe2\x80\x8babc") == b"\xe2\x80\x8babc"               # U+200B is not white space
    assert ts(b"\x00abc\x00") == b"\x00abc\x00"
    assert ts(b"   ") == b""
    assert ts(b" a b ") == b"a b"
    assert ts(b"\xe2\x80\x80\x80") == b"\x80"                         # space rune, then a stray continuation byte
    assert ts(b"\xc0\x8aabc") == b"\xc0\x8aabc"                       # over-long encoding is invalid


def test_pad_field_counts_runes(oracle):
    assert oracle.pad_field(b"test-bucket", 16) == b"test-bucket     "
    assert oracle.pad_field(b"0123456789abcdefXYZ", 16) == b"0123456789abcdefXYZ"      # never truncates
    assert oracle.pad_field("café".encode(), 6) == "café".encode() + b"  "   # 4 runes, 5 bytes
    assert oracle.pad_field(b"\xff\xfe", 4) == b"\xff\xfe  "                           # invalid bytes = 1 rune each


def test_generate_errors(oracle):
    with pytest.raises(oracle.OracleError, match=r"expected: 512, actual: 514"):
        oracle.generate(512, 1, "b", "0123456789abc", 1, T0)                           # 13-byte key
    with pytest.raises(oracle.OracleError, match=r"expected: 300, actual: 256"):
        oracle.generate(300, 1, "b", "k", 1, T0)
    assert oracle.generate(0, 1, "b", "k", 1, T0) == b""
    with pytest.raises(oracle.OracleError, match=r"expected: 255, actual: 0"):
        oracle.generate(255, 1, "b", "k", 1, T0)


def test_hex_dump_layout(oracle):
    full = oracle.hex_dump(bytes(range(32)))
    lines = full.split("\n")
    assert lines[0] == "00000000  00 01 02 03 04 05 06 07  08 09 0a 0b 0c 0d 0e 0f  |................|"
    assert len(lines[0]) == 78 and full.endswith("|\n")
    assert oracle.hex_dump(b"") == ""
    part = oracle.hex_dump(b"abcdefghij")
    # the bar of the right-hand column stays in column 60 on a partial line
    assert part == "00000000  61 62 63 64 65 66 67 68  69 6a" + " " * 20 + "|abcdefghij|\n"
    assert len(part) == 61 + 10 + 2


def test_hex_dump_matches_the_go_documentation_example(oracle):
    """encoding/hex's own documented example (ExampleDump in the Go standard
    library's package documentation; not a file of the reference): 42 bytes, so the
    third line is a partial one.  The only published vector for the partial-line
    layout that the short-read message of Valid (pattern.go:126) prints."""
    want = ("00000000  47 6f 20 69 73 20 61 6e  20 6f 70 65 6e 20 73 6f  |Go is an open so|\n"
            "00000010  75 72 63 65 20 70 72 6f  67 72 61 6d 6d 69 6e 67  |urce programming|\n"
            "00000020  20 6c 61 6e 67 75 61 67  65 2e                    | language.|\n")
    assert oracle.hex_dump(b"Go is an open source programming language.") == want


def test_short_read_messages_assembled_from_the_readme_layout(oracle, golden):
    """Valid over a body that ends early (pattern.go:124-127).  The expected texts
    are put together by tests/golden/make_golden.py from what the reference pins:
    whole hex.Dump lines and annotation rows as README.md:100-121 prints them, and
    dump's rule for text that runs out early (pattern.go:195-199; 0, 16 and 32
    bytes read).  For the 700-byte body (188 bytes of the third unit) everything up
    to the last whole line is pinned that way; the final twelve-byte line is the
    part no file of the reference shows (`unpinned_tail`)."""
    g = golden["short_read"]
    st = g["state"]
    for c in g["cases"]:
        body = bytes.fromhex(c["body_hex"])
        kind, unit, text = oracle.valid(st["worker_id"], st["bucket"], st["key"], c["size"], st["write_count"], body)
        assert (kind, unit) == (2, 2), c["read"]
        assert text.startswith(c["message"]), c["read"]
        assert text == c["message"] + c["unpinned_tail"], c["read"]
    assert [c["read"] for c in g["cases"]] == [0, 16, 32, 176, 188] and len(bytes.fromhex(g["cases"][-1]["body_hex"])) == 700


def test_dump_annotations(oracle):
    assert oracle.dump("") == ""
    text = oracle.hex_dump(bytes(256))
    out = oracle.dump(text).split("\n")
    assert out[1].endswith("bucket name") and out[3].endswith("key name") and out[4].endswith("write count")
    assert out[6].endswith("byte offset in this object") and out[7].endswith("worker ID")
    assert out[8].endswith("unix time (micro sec)")
    assert len(out) == 16 + 6 + 1
    short = oracle.dump(oracle.hex_dump(b"xy"))                   # pattern.go:195-199
    assert short.count("\n") == 2 and short.endswith("bucket name\n")


def test_valid_negative_cases(oracle):
    b, k, wc, wid = "test-bucket", "ov0000000001", 7, 0x1234
    good = bytearray(oracle.generate(1024, wid, b, k, wc, T0))
    # every reference-visible field, one at a time, in unit 2
    for off, bit in [(3, 1), (17, 2), (28, 4), (33, 8), (36, 16)]:
        bad = bytearray(good)
        bad[512 + off] ^= 0x40
        kind, unit, text = oracle.valid(wid, b, k, 1024, wc, bytes(bad))
        assert (kind, unit) == (1, 2)
        m, t2 = oracle.valid_data_unit(2, wid, b, k, wc, bytes(bad[512:768]))
        assert m == bit and t2 == text
    # time and filler are not validated (pattern.go:172)
    for off in (40, 47, 48, 100, 255):
        bad = bytearray(good)
        bad[off] ^= 0xFF
        assert oracle.valid(wid, b, k, 1024, wc, bytes(bad))[0] == 0
    # short reads: the bad unit wins over the short read; n may be 0
    assert oracle.valid(wid, b, k, 1024, wc, bytes(good[:768]))[:2] == (2, 3)
    kind, unit, text = oracle.valid(wid, b, k, 1024, wc, bytes(good[:768]))
    assert text.startswith("could not read some data. (expected: 256byte, actual: 0byte)\n")
    bad = bytearray(good[:700]); bad[256 + 30] ^= 1
    assert oracle.valid(wid, b, k, 1024, wc, bytes(bad))[:2] == (1, 1)
    # extra trailing bytes are not detected; Size 0 is always fine
    assert oracle.valid(wid, b, k, 1024, wc, bytes(good) + b"junk")[0] == 0
    assert oracle.valid(wid, b, k, 0, wc, b"")[0] == 0
    # padding that is white space but not 0x20 still passes TrimSpace
    ws = bytearray(good); ws[15] = 0x09; ws[14] = 0x0A
    assert oracle.valid(wid, b, k, 1024, wc, bytes(ws))[0] == 0
    # name shifted right by a leading space passes too
    sh = bytearray(good); sh[0:16] = b" test-bucket    "
    assert oracle.valid(wid, b, k, 1024, wc, bytes(sh))[0] == 0


def test_batch_forms_agree_with_scalar(oracle):
    rng = np.random.default_rng(5)
    D = oracle.DESC_DTYPE
    n = 7
    d = np.zeros(n, dtype=D)
    off = 0
    for i in range(n):
        key = oracle.generate_key(i).encode()
        d[i]["bucket"] = np.frombuffer(b"test-bucket     ", np.uint8)
        d[i]["bucket_len"] = 11
        d[i]["key"] = np.frombuffer(key, np.uint8)
        d[i]["key_len"] = 12
        d[i]["write_count"] = i + 1
        d[i]["worker_id"] = 100 + i
        d[i]["unix_micro"] = T0 + i
        d[i]["n_units"] = int(rng.integers(0, 6))
        d[i]["body_offset"] = off
        off += int(d[i]["n_units"]) * 256
    body = np.zeros(off, dtype=np.uint8)
    oracle.fill_batch(d, body)
    for i in range(n):
        exp = oracle.generate(int(d[i]["n_units"]) * 256, 100 + i, "test-bucket", oracle.generate_key(i), i + 1, T0 + i)
        o = int(d[i]["body_offset"])
        assert body[o:o + len(exp)].tobytes() == exp
    res, cnt = oracle.validate_batch(d, body)
    assert (res["first_bad_unit"] == oracle.NO_BAD).all() and cnt[1] == 0 and cnt[0] == n
    assert cnt[2] == d["n_units"].sum()
