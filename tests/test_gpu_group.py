"""KV -> KMV on the device (mgs_group_device) against the CPU oracle's
restatement of KMVContainer::convert (src/kmvcontainer.h:74-150): unique keys
in the order of first occurrence (HashBucket's entry order,
src/hashbucket.h:263-342), every key's values in stream order, value buffers
closed at page_bytes.  Index and byte work: exact.

The oracle for this row restates the reference; it is not pinned on compiled
reference code (HashBucket / KMVContainer need the reference's memory.cpp and
its configure-generated header).  test_oracle.py checks it against an
independent grouping written with Python dicts."""
import numpy as np
import pytest

from mimir_amd import capi, datagen
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def check(oracle, blob, offs, lay_t, n, page_bytes=64 << 20, hashscale=1):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mimir_amd.packer import DevicePacker, group_records
    glay, olay = capi.Layout.make(*lay_t), O.Layout.make(*lay_t)
    want = oracle.kmv_convert(olay, blob, offs, n, hashscale=hashscale, page_bytes=page_bytes)
    pk = DevicePacker(glay, 2, -1, 4096, 1)
    d = torch.from_numpy(np.ascontiguousarray(blob)).cuda() if len(blob) else torch.empty(1, dtype=torch.uint8, device="cuda")
    d_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).cuda() if offs is not None else None
    first, start, values, cuts = group_records(pk, d, n, rec_off=d_off, page_bytes=page_bytes, max_cuts=4096)
    assert np.array_equal(first.cpu().numpy().view(np.uint32), want[0])
    assert np.array_equal(start.cpu().numpy().view(np.uint32).astype(np.uint64), want[1])
    assert np.array_equal(values.cpu().numpy(), want[2])
    assert cuts == want[3].tolist()
    pk.close()
    return len(want[0]), len(cuts)


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 2047, 2048, 2049, 50000])
def test_group_words_sizes(oracle, n):
    blob, offs, lay = datagen.word_records(max(n, 1), n_unique=97, alpha=0.9, len_mean=5, len_sd=2, seed=n + 5)
    if n == 0:
        blob, offs = blob[:0], offs[:1]
    check(oracle, blob, offs, lay, n, page_bytes=4096)


def test_group_words_many_keys_and_buffers(oracle):
    n = 2_000_000
    blob, offs, lay = datagen.word_records(n, n_unique=300000, alpha=1.0, len_mean=7, len_sd=2, seed=21)
    groups, ncuts = check(oracle, blob, offs, lay, n, page_bytes=1 << 20)
    assert groups > 50000 and ncuts >= 10


def test_group_all_distinct_and_all_equal(oracle):
    n = 300000
    blob, _, lay = datagen.u64_records(n, seed=8)  # distinct keys: every record its own group
    groups, _ = check(oracle, blob, None, lay, n, page_bytes=1 << 16)
    assert groups == n
    same = np.tile(np.frombuffer(b"samekey\0" + (7).to_bytes(8, "little"), dtype=np.uint8), n)
    view = same.reshape(n, 16).copy()
    view[:, 8:] = np.arange(n, dtype=np.uint64).view(np.uint8).reshape(n, 8)  # values tell the order
    offs = np.arange(n + 1, dtype=np.uint64) * 16
    groups, _ = check(oracle, view.reshape(-1), offs, (capi.KEY_CSTR, 0, capi.VAL_FIXED, 8), n, page_bytes=1 << 16)
    assert groups == 1


def test_group_other_layouts(oracle):
    n = 200000
    blob, offs, lay = datagen.join_records(n, n_unique=3000, alpha=1.0, len_mean=6, len_sd=1, seed=45)
    check(oracle, blob, offs, lay, n, page_bytes=1 << 18, hashscale=8)  # 16-byte values
    blob, _, lay = datagen.octree_records(n, level=6, seed=3)  # fixed 6-char keys, odd record width
    groups, _ = check(oracle, blob, None, lay, n, page_bytes=1 << 18)
    assert groups < n
    rng = np.random.default_rng(4)  # keys only (no value)
    keys = [bytes(rng.integers(97, 100, size=rng.integers(0, 4), dtype=np.uint8)) + b"\0" for _ in range(n // 4)]
    blob = np.frombuffer(b"".join(keys), dtype=np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(k) for k in keys])]).astype(np.uint64)
    check(oracle, blob, offs, (capi.KEY_CSTR, 0, capi.VAL_VOID, 0), len(keys))


@pytest.mark.parametrize("vbytes", [1, 4, 7, 8, 9, 12, 24])
def test_group_value_widths(oracle, vbytes):
    """Values narrower than, equal to and wider than the 8 bytes that travel inside the sort
    records (wider ones fetch their tail from the record), behind string keys of 1..20 bytes:
    keys longer than 8 bytes share their first 8 bytes with others (the slot's copy of a key's
    first bytes decides nothing then, the stream does)."""
    n = 120_000
    rng = np.random.default_rng(vbytes)
    stems = [b"k", b"abcdefg", b"abcdefgh", b"abcdefghi", b"abcdefghijklmnopqrs", b"abcdefghX", b"abcdefgh1", b"zz"]
    keys = [stems[int(j)] + bytes([97 + int(t)]) * int(r) + b"\0"
            for j, t, r in zip(rng.integers(0, len(stems), n), rng.integers(0, 3, n), rng.integers(0, 2, n))]
    vals = rng.integers(0, 256, size=(n, vbytes), dtype=np.uint8)
    recs = [k + v.tobytes() for k, v in zip(keys, vals)]
    blob = np.frombuffer(b"".join(recs), dtype=np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint64)
    groups, _ = check(oracle, blob, offs, (capi.KEY_CSTR, 0, capi.VAL_FIXED, vbytes), n, page_bytes=1 << 16)
    assert groups == len(set(keys))


def test_group_keys_that_share_a_prefix_and_a_hash_slot(oracle):
    """Keys differing only in length or in the last byte, many of them colliding in the table."""
    n = 100000
    rng = np.random.default_rng(9)
    stems = [b"a" * k for k in range(1, 40)] + [b"a" * 20 + bytes([c]) for c in range(98, 123)]
    keys = [stems[i] + b"\0" for i in rng.integers(0, len(stems), size=n)]
    recs = [k + int(i).to_bytes(8, "little") for i, k in enumerate(keys)]
    blob = np.frombuffer(b"".join(recs), dtype=np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint64)
    groups, _ = check(oracle, blob, offs, (capi.KEY_CSTR, 0, capi.VAL_FIXED, 8), n, page_bytes=50000)
    assert groups == len(stems)


def test_group_errors(oracle):
    import torch
    from mimir_amd.packer import DevicePacker, group_records
    blob, offs, lay = datagen.word_records(1000, n_unique=500, alpha=0.5, len_mean=5, len_sd=2, seed=2)
    pk = DevicePacker(capi.Layout.make(*lay), 2, -1, 4096, 1)
    d = torch.from_numpy(blob).cuda()
    d_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).cuda()
    with pytest.raises(capi.MgsError) as ei:  # more groups than room
        group_records(pk, d, 1000, rec_off=d_off, group_capacity=10)
    assert ei.value.status == capi.ERR_ARG
    with pytest.raises(capi.MgsError) as ei:  # more cuts than room
        group_records(pk, d, 1000, rec_off=d_off, page_bytes=64, max_cuts=2)
    assert ei.value.status == capi.ERR_ARG
    bad = blob.copy()
    bad[int(offs[500]) - 9] = 65  # the key's NUL is gone
    with pytest.raises(capi.MgsError) as ei:
        group_records(pk, torch.from_numpy(bad).cuda(), 1000, rec_off=d_off)
    assert ei.value.status == capi.ERR_MALFORMED
    pk.close()
    pk = DevicePacker(capi.Layout.make(capi.KEY_CSTR, 0, capi.VAL_CSTR, 0), 2, -1, 4096, 1)
    with pytest.raises(capi.MgsError) as ei:
        group_records(pk, d, 1000, rec_off=d_off)
    assert ei.value.status == capi.ERR_ARG
    pk.close()
