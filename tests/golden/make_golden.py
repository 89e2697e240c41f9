#!/usr/bin/env python3
"""Extracts the reference's known-answer data into JSON fixtures.

Run once in the build container (needs /root/reference); the outputs are
committed so that no test reads the reference tree at run time.

Sources (data only, no reference source text is copied):
  * README.md:97-122 — console output of a real run: one mismatch message and
    the annotated hex dump of one complete 256-byte unit.
  * internal/pattern/pattern_test.go:30-255 — the byte values and size ranges
    its assertions state, typed in below as numbers/strings.
  * short_read.json — the message of Valid for a body that ends early
    (pattern.go:126), assembled from what the reference itself pins: the layout
    of a full hex.Dump line and the three annotation rows as README.md:100-108
    prints them, and dump's rule for text that runs out before the annotations
    do (pattern.go:195-199).  The one thing no file of the reference shows, a
    hex.Dump line of fewer than 16 bytes, is kept apart as `unpinned_tail`.
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def readme_fixture():
    lines = open(os.path.join(REF, "README.md"), encoding="utf-8").read().split("\n")
    # README.md lines are 1-based; message = lines 98..121
    msg_lines = lines[97:121]
    assert msg_lines[0].startswith("- WriteCount is wrong."), msg_lines[0]
    assert msg_lines[-1].startswith("000000f0"), msg_lines[-1]
    message = "\n".join(msg_lines) + "\n"
    unit = bytearray()
    for ln in msg_lines:
        m = re.match(r"^([0-9a-f]{8})  ((?:[0-9a-f]{2} +){16})\|", ln)
        if m:
            assert int(m.group(1), 16) == len(unit)
            unit += bytes(int(x, 16) for x in m.group(2).split())
    assert len(unit) == 256
    return {
        "source": "README.md:97-122",
        "unit_hex": unit.hex(),
        # state that produced the unit (read off its own bytes)
        "written": {
            "bucket": "test-bucket", "key": "ov00020000e2", "write_count": 1,
            "unit_count": 1, "worker_id": 0x561B,
            "unix_micro": int.from_bytes(unit[40:48], "little"),
        },
        # state the validator expected when it printed the message (README.md:98-99,122)
        "expected_state": {
            "bucket": "test-bucket", "key": "ov00020000e2", "write_count": 2,
            "unit_count": 0, "worker_id": 0x561B,
        },
        "message": message,
        "caller_prefix": "data validation error occurred after put.\n",
    }


def pattern_test_fixture():
    bucket, long_bucket, key = "test-bucket", "test-long-long-long-bucket", "test-key"
    hdr = lambda off: {  # noqa: E731
        "write_count_bytes": "2c010000", "offset_bytes": off, "worker_bytes": "64000000"}
    return {
        "source": "internal/pattern/pattern_test.go:30-255",
        "obj": {"key": key, "write_count": 300}, "worker_id": 100,
        "bucket": bucket, "long_bucket": long_bucket,
        "bucket_field_hex": (bucket.encode() + b" " * 5).hex(),
        "long_bucket_field_hex": long_bucket.encode()[:16].hex(),
        "key_field_hex": (key.encode() + b" " * 4).hex(),
        "generate_data_unit": {"unit_count": 4, "len": 256, **hdr("00040000")},      # :30-57
        "generate_512": [hdr("00000000"), hdr("00010000")],                          # :59-106
        "generate_long_bucket_512": [hdr("00000000"), hdr("00010000")],              # :108-155
        "valid_round_trips": [                                                       # :157-200
            {"what": "unit", "unit_count": 4, "bucket": bucket},
            {"what": "object", "size": 1024, "bucket": bucket},
            {"what": "object", "size": 1024, "bucket": long_bucket},
        ],
        "decide_size": [                                                             # :202-255
            {"min": 512, "max": 512, "err": False},
            {"min": 1024, "max": 5 * 1024 * 1024, "err": False},
            {"min": 256, "max": 1024, "err": False},
            {"min": 513, "max": 1024, "err": True},
            {"min": 512, "max": 513, "err": True},
            {"min": 0, "max": 512, "err": True},
            {"min": 1024, "max": 512, "err": True},
        ],
    }


def short_read_fixture():
    rd = readme_fixture()
    unit1 = bytes.fromhex(rd["unit_hex"])
    dump_lines = [ln for ln in rd["message"].split("\n") if re.match(r"^[0-9a-f]{8}  ", ln)]
    notes = [ln for ln in rd["message"].split("\n") if ln.startswith("   ")]
    assert len(dump_lines) == 16 and len(notes) == 6
    ann = ["\n".join(notes[0:1]) + "\n", "\n".join(notes[1:3]) + "\n", "\n".join(notes[3:6]) + "\n"]

    def full_line(off, b):
        # a complete line as README.md:100-121 prints sixteen of them
        assert len(b) == 16
        hx = " ".join("%02x" % x for x in b[:8]) + "  " + " ".join("%02x" % x for x in b[8:])
        tx = "".join(chr(x) if 0x20 <= x <= 0x7e else "." for x in b)
        return "%08x  %s  |%s|" % (off, hx, tx)

    for i, ln in enumerate(dump_lines):                       # the formatter is held to all sixteen README lines
        assert full_line(16 * i, unit1[16 * i:16 * i + 16]) == ln, (i, ln)

    def unit(k):                                              # the README unit at another place in its object
        u = bytearray(unit1)
        u[32:36] = (k * 256).to_bytes(4, "little")
        return bytes(u)

    def dump_of_full_lines(data):                             # pattern.go:182-206 over whole lines only
        text = [full_line(o, data[o:o + 16]) + "\n" for o in range(0, len(data), 16)]
        out = ""
        for a in ann:
            if not text:                                      # pattern.go:195-199: nothing left, the annotation still goes out
                return out + a
            out += text.pop(0) + a
        return out + "".join(text)

    cases = []
    for n in (0, 16, 32, 176):
        body = unit(0) + unit(1) + unit(2)[:n]
        msg = "could not read some data. (expected: 256byte, actual: %dbyte)\n" % n
        msg += dump_of_full_lines(unit(2)[:n]) if n else ""
        cases.append({"body_hex": body.hex(), "size": 768, "read": n, "message": msg, "unpinned_tail": ""})
    # 700 bytes: 188 of unit 2 arrive = eleven whole lines and one of twelve bytes.  The whole lines and
    # annotations above are pinned; the twelve-byte line follows encoding/hex.Dump as Go documents it (hex
    # columns padded with spaces to full width, then the characters that exist) and is NOT pinned by any
    # file of the reference.
    n = 188
    body = unit(0) + unit(1) + unit(2)[:n]
    pinned = "could not read some data. (expected: 256byte, actual: %dbyte)\n" % n + dump_of_full_lines(unit(2)[:176])
    last = unit(2)[176:188]
    hx = " ".join("%02x" % x for x in last[:8]) + "  " + " ".join("%02x" % x for x in last[8:])
    tail = "%08x  %s" % (176, hx) + " " * (3 * 4) + "  |" + "".join(chr(x) if 0x20 <= x <= 0x7e else "." for x in last) + "|\n"
    cases.append({"body_hex": body.hex(), "size": 768, "read": n, "message": pinned, "unpinned_tail": tail})
    return {
        "source": "README.md:100-121 (line layout, annotation rows) + internal/pattern/pattern.go:126,182-206",
        "state": {"bucket": "test-bucket", "key": "ov00020000e2", "write_count": 1, "worker_id": 0x561B},
        "cases": cases,
    }


if __name__ == "__main__":
    json.dump(readme_fixture(), open(os.path.join(HERE, "readme_unit.json"), "w"), indent=1)
    json.dump(pattern_test_fixture(), open(os.path.join(HERE, "pattern_test.json"), "w"), indent=1)
    json.dump(short_read_fixture(), open(os.path.join(HERE, "short_read.json"), "w"), indent=1)
    print("wrote readme_unit.json, pattern_test.json, short_read.json")
