import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this process")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O


@pytest.fixture(scope="session")
def golden():
    import json

    d = os.path.join(ROOT, "tests", "golden")
    return {
        "readme": json.load(open(os.path.join(d, "readme_unit.json"))),
        "pattern_test": json.load(open(os.path.join(d, "pattern_test.json"))),
        "short_read": json.load(open(os.path.join(d, "short_read.json"))),
    }


def session_small_object_bytes():
    """OVAL_TEST_SMALL_OBJECT_BYTES=N in the environment runs the whole suite with
    the small-object kernels selected for layouts that average under N bytes per
    object (2**64 - 1: every layout), to hold them to every parity test there is."""
    from oval_amd import engine

    v = os.environ.get("OVAL_TEST_SMALL_OBJECT_BYTES")
    return (int(v, 0), int(v, 0)) if v else engine.SMALL_OBJECT_BYTES_DEFAULT


def session_big_object_bytes():
    """OVAL_TEST_BIG_OBJECT_BYTES=N likewise for the validate kernel for multipart-sized
    bodies (layouts that average at least N bytes per object; 0: every layout)."""
    from oval_amd import engine

    v = os.environ.get("OVAL_TEST_BIG_OBJECT_BYTES")
    return int(v, 0) if v else engine.BIG_OBJECT_BYTES_DEFAULT


@pytest.fixture(scope="session", autouse=True)
def _small_object_selection():
    if os.environ.get("OVAL_TEST_SMALL_OBJECT_BYTES") or os.environ.get("OVAL_TEST_BIG_OBJECT_BYTES"):
        from oval_amd import engine

        engine.set_small_object_bytes(*session_small_object_bytes())
        engine.set_big_object_bytes(session_big_object_bytes())
    yield
