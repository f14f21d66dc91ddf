"""The host-staged front end through the C ABI (ctypes): the two-call pack of the
non-blocking shuffler (mgs_stager_pack_begin / _end) gives what the one-call pack gives,
a round's partitions copied straight into a caller's page-locked buffer
(mgs_stager_send_image_to, mgs_host_alloc) are the bytes of the stager's own host view, and
calls out of order are refused.  Checked against the oracle's round
(CollectiveShuffler::write, src/collectiveshuffler.h:131-166).  Byte work: exact."""
import ctypes as C

import numpy as np
import pytest

from mimir_amd import capi
from oracle import oracle as O

pytestmark = pytest.mark.gpu
P, BUF = 4, 1 << 16


def _make(lib, rank):
    d = capi.Desc()
    d.nranks, d.rank, d.buf_size, d.device = P, rank, BUF, 0
    d.layout = capi.Layout.make(capi.KEY_FIXED, 8, capi.VAL_FIXED, 8)
    d.max_records = 1 << 20
    h = C.c_void_p()
    assert lib.mgs_create(C.byref(d), C.byref(h)) == capi.OK, lib.mgs_last_error(None)
    s = C.c_void_p()
    assert lib.mgs_stager_create(h, 1 << 20, C.byref(s)) == capi.OK
    return h, s


def _put_all(lib, s, recs):
    for i in range(len(recs) // 16):
        k = recs[16 * i:16 * i + 8].tobytes()
        v = recs[16 * i + 8:16 * i + 16].tobytes()
        assert lib.mgs_stager_put(s, k, 8, v, 8) == capi.OK


@pytest.mark.parametrize("rank", [-1, 2])
def test_pack_in_two_calls_and_image_into_caller_buffer(oracle, rank):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    lib = capi.load()
    rng = np.random.default_rng(7 + rank)
    n = 3000  # fits one round: 3000 * 16 / 4 partitions is below 64 KiB each
    recs = rng.integers(0, 256, size=16 * n, dtype=np.uint8)
    want = oracle.pack_round(O.Layout.make(capi.KEY_FIXED, 8, capi.VAL_FIXED, 8), recs, None, n, 0, P, rank, BUF)
    assert want["next"] == n
    h1, s1 = _make(lib, rank)
    h2, s2 = _make(lib, rank)
    _put_all(lib, s1, recs)
    _put_all(lib, s2, recs)
    r1, r2 = capi.Round(), capi.Round()
    loc1, loc2 = C.c_void_p(), C.c_void_p()
    dr1, dr2 = C.c_int(-1), C.c_int(-1)
    # out of order: the second half without the first, the first half twice
    assert lib.mgs_stager_pack_end(s1, C.byref(r1), C.byref(loc1), C.byref(dr1)) == capi.ERR_STATE
    assert lib.mgs_stager_pack_begin(s1) == capi.OK
    assert lib.mgs_stager_pack_begin(s1) == capi.ERR_STATE
    # nothing else between the two halves: the image is being written
    img0, so0 = C.c_void_p(), C.POINTER(C.c_int64)()
    assert lib.mgs_stager_send_image_host(s1, C.byref(img0), C.byref(so0)) == capi.ERR_STATE
    assert lib.mgs_stager_round_done(s1) == capi.ERR_STATE
    assert lib.mgs_stager_put(s1, b"12345678", 8, b"12345678", 8) == capi.ERR_STATE
    assert lib.mgs_stager_pack_end(s1, C.byref(r1), C.byref(loc1), C.byref(dr1)) == capi.OK
    assert lib.mgs_stager_pack(s2, C.byref(r2), C.byref(loc2), C.byref(dr2)) == capi.OK
    assert dr1.value == 1 and dr2.value == 1
    for r in (r1, r2):
        assert r.next == n and r.packed_records + r.local_records == n
        assert [r.send_offset[p] for p in range(P)] == want["send_offset"].tolist()
        assert r.local_bytes == len(want["local"])
    if rank >= 0:
        for loc in (loc1, loc2):
            got = np.frombuffer(C.string_at(loc.value, len(want["local"])), dtype=np.uint8)
            assert np.array_equal(got, want["local"])
    # the round's partitions: the stager's own host view against a caller's page-locked buffer
    img, so = C.c_void_p(), C.POINTER(C.c_int64)()
    assert lib.mgs_stager_send_image_host(s2, C.byref(img), C.byref(so)) == capi.OK
    mine = lib.mgs_host_alloc(P * BUF)
    assert mine
    so1 = C.POINTER(C.c_int64)()
    assert lib.mgs_stager_send_image_to(s1, mine, C.byref(so1)) == capi.OK
    for p in range(P):
        nb = int(want["send_offset"][p])
        assert so[p] == nb and so1[p] == nb
        a = np.frombuffer(C.string_at(img.value + p * BUF, nb), dtype=np.uint8)
        b = np.frombuffer(C.string_at(mine + p * BUF, nb), dtype=np.uint8)
        assert np.array_equal(a, want["partitions"][p]) and np.array_equal(b, want["partitions"][p])
    lib.mgs_host_free(mine)
    # an empty stager: both halves succeed and report a drained, empty round
    assert lib.mgs_stager_round_done(s1) == capi.OK
    assert lib.mgs_stager_pack_begin(s1) == capi.OK
    assert lib.mgs_stager_pack_end(s1, C.byref(r1), C.byref(loc1), C.byref(dr1)) == capi.OK
    assert dr1.value == 1 and r1.packed_records == 0 and all(r1.send_offset[p] == 0 for p in range(P))
    for h, s in ((h1, s1), (h2, s2)):
        lib.mgs_stager_destroy(s)
        lib.mgs_destroy(h)
