import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The reference's own compiled hash.cpp/serializer.h, where available."""
    from oracle import oracle as o
    try:
        return o.Reference()
    except (FileNotFoundError, OSError):
        pytest.skip("oracle/_ref/libmimir_ref.so not built (no /root/reference)")


@pytest.fixture(scope="session")
def reference_serializer():
    """The reference's own Serializer<K,V>::kv_to_bytes / get_hash_code, where available."""
    from oracle import oracle as o
    try:
        return o.ReferenceSerializer()
    except (FileNotFoundError, OSError):
        pytest.skip("oracle/_ref/libmimir_ref_ser.so not built (no /root/reference or no MPI library)")
