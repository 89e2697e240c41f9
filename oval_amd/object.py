"""Mirror of the reference's internal/object package (object.go:13-50): the
expected state of one object and the key naming scheme."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

from . import _lib

MaxBucketNameLength = 16  # object.go:14
MaxKeyLength = 12         # object.go:15
KeyShortPrefix = "ov"     # object.go:16
KeyPrefixLength = 6       # object.go:18


@dataclass
class Object:
    """object.go:21-25"""
    Key: str = ""
    Size: int = 0
    WriteCount: int = 0

    def Clear(self) -> None:  # object.go:35-38
        self.Size = 0
        self.WriteCount = 0


def generateKey(objID: int) -> str:
    """object.go:48-50"""
    buf = C.create_string_buffer(48)
    n = _lib.lib().oval_object_generate_key(objID, buf, 48)
    return buf.raw[:n].decode()


def NewObject(objID: int) -> Object:
    """object.go:40-46"""
    return Object(Key=generateKey(objID), Size=0, WriteCount=0)
