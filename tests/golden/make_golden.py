#!/usr/bin/env python3
"""Extracts the reference's known-answer data into JSON fixtures.

Run once in the build container (needs /root/reference); the outputs are
committed so that no test reads the reference tree at run time.

Sources (data only, no reference source text is copied):
  * README.md:97-122 — console output of a real run: one mismatch message and
    the annotated hex dump of one complete 256-byte unit.
  * internal/pattern/pattern_test.go:30-255 — the byte values and size ranges
    its assertions state, typed in below as numbers/strings.
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


if __name__ == "__main__":
    json.dump(readme_fixture(), open(os.path.join(HERE, "readme_unit.json"), "w"), indent=1)
    json.dump(pattern_test_fixture(), open(os.path.join(HERE, "pattern_test.json"), "w"), indent=1)
    print("wrote readme_unit.json, pattern_test.json")
