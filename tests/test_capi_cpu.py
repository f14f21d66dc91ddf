"""CPU-side checks of the C ABI: the library loads, exports every symbol the
header declares, and its host-only entry points agree with the oracle.  No
kernel is launched here."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from mimir_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mimir_gpu_shuffle.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mgs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    names = _declared_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in capi.SYMBOLS, f"{n} has no ctypes signature in capi.SYMBOLS"
    assert sorted(capi.SYMBOLS) == names


def test_struct_sizes_match_header():
    # mgs_layout 6 x int32, mgs_desc, mgs_round as laid out in the header
    assert C.sizeof(capi.Layout) == 24
    assert C.sizeof(capi.Desc) == 4 + 4 + 24 + 8 + 4 + 4 + 8
    assert C.sizeof(capi.Round) == 32 + 8 * capi.MGS_MAX_RANKS


def test_geometry_matches_oracle(oracle):
    for comm in (1000, 8192, 64 << 10, 1 << 20, 64 << 20, 1 << 30, 4 << 30):
        for P in (1, 2, 3, 4, 5, 7, 8, 16, 64, 200):
            try:
                want = oracle.geometry(comm, P)
            except ValueError:
                with pytest.raises(capi.MgsError):
                    capi.geometry(comm, P)
                continue
            assert capi.geometry(comm, P) == want


def test_host_hash_matches_golden_and_oracle(oracle):
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "hashlittle_vectors.json")))
    for v in g["initval0"]:
        assert capi.hashlittle(bytes.fromhex(v["key_hex"])) == v["hash"]
    for v in g["seeded"]:
        assert capi.hashlittle(bytes.fromhex(v["key_hex"]), v["initval"]) == v["hash"]
    rng = np.random.default_rng(3)
    for L in range(0, 70):
        key = rng.integers(0, 256, size=L, dtype=np.uint8).tobytes()
        assert capi.hashlittle(key) == oracle.hashlittle(key)


def test_division_free_modulo_is_exact():
    """dest = hash % (uint32_t) nranks (src/baseshuffler.h:277) for every
    divisor the library accepts, on edge values and a random sweep."""
    lib = capi.load()
    rng = np.random.default_rng(11)
    edge = [0, 1, 2, 3, 0x7fffffff, 0x80000000, 0x80000001, 0xfffffffe, 0xffffffff]
    rnd = rng.integers(0, 1 << 32, size=400, dtype=np.uint64).tolist()
    for P in range(1, capi.MGS_MAX_RANKS + 1):
        mult = [k * P + o for k in (1, 2, (0xffffffff // P)) for o in (-1, 0, 1)]
        for x in edge + rnd[:60] + mult:
            x &= 0xffffffff
            assert lib.mgs_dest_of_hash_host(x, P) == x % P, (x, P)
    for P in (3, 5, 6, 7, 24, 63, 255):
        for x in rnd:
            assert lib.mgs_dest_of_hash_host(int(x), P) == int(x) % P


def test_create_rejects_bad_descriptors():
    lib = capi.load()
    lay = capi.Layout.make(capi.KEY_FIXED, 8, capi.VAL_FIXED, 8)
    h = C.c_void_p()
    for nranks, rank, buf in [(0, 0, 64), (capi.MGS_MAX_RANKS + 1, 0, 64), (4, 4, 64), (4, 0, 0)]:
        d = capi.Desc(nranks, rank, lay, buf, 0, 0, 1024)
        assert lib.mgs_create(C.byref(d), C.byref(h)) == capi.ERR_ARG
        assert lib.mgs_last_error(None)
    bad = capi.Layout.make(capi.KEY_FIXED, 0, capi.VAL_FIXED, 8)
    d = capi.Desc(4, 0, bad, 64, 0, 0, 1024)
    assert lib.mgs_create(C.byref(d), C.byref(h)) == capi.ERR_ARG
