"""The send side with a combiner (mgs_combine_device) against the CPU oracle's
restatement of CombineCollectiveShuffler::write
(src/combinecollectiveshuffler.h:63-180) with a declared operation in place of
the user callback: per round, the first record of every key in its partition
in emit order with the combined value, local records untouched, the round cut
before the first new key that does not fit.  Byte work: exact.

Parity unpinned: the reference class needs HashBucket and the reference's
memory.cpp (configure-generated header), so the oracle is a restatement;
test_oracle.py checks it through word counts."""
import numpy as np
import pytest

from mimir_amd import capi, datagen
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def run_rounds(oracle, blob, offs, lay_t, n, P, rank, buf, op=capi.COMBINE_SUM_U64):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mimir_amd.packer import DevicePacker
    glay, olay = capi.Layout.make(*lay_t), O.Layout.make(*lay_t)
    pk = DevicePacker(glay, P, rank, buf, max(n, 1))
    d = torch.from_numpy(np.ascontiguousarray(blob)).cuda() if len(blob) else torch.empty(1, dtype=torch.uint8, device="cuda")
    d_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).cuda() if offs is not None else None
    image = pk.new_send_image()
    local = torch.empty(max(len(blob), 16), dtype=torch.uint8, device="cuda")
    start, rounds = 0, 0
    while True:
        want = oracle.combine_round(olay, blob, offs, n, start, P, rank, buf, op)
        got = pk.combine(d, n, image, start=start, rec_off=d_off, local_out=local, op=op)
        assert got.next == want["next"], (rounds, got.next, want["next"])
        assert got.send_offset.tolist() == want["send_offset"].tolist()
        assert (got.local_bytes, got.local_records, got.packed_records) == \
            (want["local_bytes"], want["local_records"], want["packed_records"])
        img = image.cpu().numpy()
        for p in range(P):
            so = int(want["send_offset"][p])
            assert np.array_equal(img[p * buf:p * buf + so], want["partitions"][p]), (rounds, p)
        assert np.array_equal(local[:got.local_bytes].cpu().numpy(), want["local"])
        rounds += 1
        start = got.next
        if start >= n:
            break
        assert rounds < 10000
    pk.close()
    return rounds


@pytest.mark.parametrize("P,rank", [(1, -1), (2, 0), (4, 3), (8, -1), (64, 17)])
def test_combine_words_one_round(oracle, P, rank):
    n = 60000
    blob, offs, lay = datagen.word_records(n, n_unique=700, alpha=0.9, len_mean=5, len_sd=2, seed=P)
    assert run_rounds(oracle, blob, offs, lay, n, P, rank, buf=1 << 16) == 1


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 2047, 2048, 2049])
def test_combine_ragged_sizes(oracle, n):
    blob, offs, lay = datagen.word_records(max(n, 1), n_unique=40, alpha=0.5, len_mean=4, len_sd=1, seed=n)
    if n == 0:
        blob, offs = blob[:0], offs[:1]
    run_rounds(oracle, blob, offs, lay, n, 4, 1, buf=4096)


def test_combine_round_cuts(oracle):
    """Small partitions: many rounds, keys reappear in later rounds with partial sums."""
    n = 40000
    blob, offs, lay = datagen.word_records(n, n_unique=3000, alpha=0.8, len_mean=6, len_sd=2, seed=5)
    rounds = run_rounds(oracle, blob, offs, lay, n, 4, 2, buf=2048)
    assert rounds >= 5
    rounds = run_rounds(oracle, blob, offs, lay, n, 3, -1, buf=600)
    assert rounds >= 20


@pytest.mark.parametrize("op", [capi.COMBINE_MIN_U64, capi.COMBINE_MAX_U64])
def test_combine_min_max(oracle, op):
    n = 50000
    blob, offs, lay = datagen.word_records(n, n_unique=300, alpha=0.7, len_mean=5, len_sd=1, seed=8)
    blob = blob.copy()
    ends = offs[1:].astype(np.int64)
    vals = np.random.default_rng(1).integers(0, 2**63, size=n, dtype=np.uint64)
    for k in range(8):
        blob[ends - 8 + k] = (vals >> np.uint64(8 * k)).astype(np.uint8)
    run_rounds(oracle, blob, offs, lay, n, 4, 0, buf=1 << 15, op=op)


def test_combine_fixed_keys(oracle):
    """octree-style fixed keys of odd width and <u64,u64> records."""
    n = 100000
    blob, _, lay = datagen.octree_records(n, level=5, seed=3)
    run_rounds(oracle, blob, None, lay, n, 8, 5, buf=1 << 12)
    assert run_rounds(oracle, blob, None, lay, n, 8, 5, buf=208) >= 2
    rng = np.random.default_rng(2)
    keys = rng.integers(0, 5000, size=n, dtype=np.uint64)
    recs = np.stack([keys, np.arange(n, dtype=np.uint64)], axis=1).view(np.uint8).reshape(-1)
    run_rounds(oracle, recs, None, (capi.KEY_FIXED, 8, capi.VAL_FIXED, 8), n, 8, -1, buf=1 << 14)


def test_combine_large(oracle):
    n = 3_000_000
    blob, offs, lay = datagen.word_records(n, n_unique=200000, alpha=1.0, len_mean=7, len_sd=2, seed=21)
    run_rounds(oracle, blob, offs, lay, n, 8, 3, buf=1 << 20)


def test_combine_errors(oracle):
    import torch
    from mimir_amd.packer import DevicePacker
    blob, offs, lay = datagen.word_records(1000, n_unique=500, alpha=0.5, len_mean=5, len_sd=2, seed=2)
    d = torch.from_numpy(blob).cuda()
    d_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).cuda()
    pk = DevicePacker(capi.Layout.make(*lay), 2, 0, 8, 1000)  # no record fits an 8-byte partition
    image = pk.new_send_image()
    local = torch.empty(len(blob), dtype=torch.uint8, device="cuda")
    with pytest.raises(capi.MgsError) as ei:
        pk.combine(d, 1000, image, rec_off=d_off, local_out=local)
    assert ei.value.status == capi.ERR_RECORD_SIZE
    pk.close()
    pk = DevicePacker(capi.Layout.make(*lay), 2, 0, 1 << 16, 1000)
    image = pk.new_send_image()
    with pytest.raises(capi.MgsError) as ei:  # local stream does not fit
        pk.combine(d, 1000, image, rec_off=d_off, local_out=local[:64])
    assert ei.value.status == capi.ERR_LOCAL_OVERFLOW
    pk.close()
    pk = DevicePacker(capi.Layout.make(capi.KEY_CSTR, 0, capi.VAL_FIXED, 16), 2, 0, 1 << 16, 1000)
    with pytest.raises(capi.MgsError) as ei:  # 16-byte value: no declared operation applies
        pk.combine(d, 1000, pk.new_send_image(), rec_off=d_off, local_out=local)
    assert ei.value.status == capi.ERR_ARG
    pk.close()
