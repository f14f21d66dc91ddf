"""Parity of the HIP path (through the C ABI) against the CPU oracle:
same destination for every record, byte-identical partitions in stable emit
order, same round boundaries.  Bit-exact: this is integer/byte work."""
import os

import numpy as np
import pytest

from mimir_amd import capi, datagen
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _layouts(t):
    return (capi.Layout.make(t[0], t[1], t[2], t[3]),
            O.Layout.make(t[0], t[1], t[2], t[3]))


def run_and_compare(oracle, blob, offs, lay_t, n, P, rank, buf_size, algo,
                    start=0, init=None, max_records=None, check_dest=True):
    """One pack call on the GPU vs one oracle round.  Returns the GPU result.
    Layouts the byte-granular one-sweep kernels take are run through both of them: the
    LDS-image kernel (the default for a one-round call) and the round-cutting kernel."""
    first = _run_and_compare(oracle, blob, offs, lay_t, n, P, rank, buf_size, algo, start, init, max_records,
                             check_dest, 0)
    if first[1] == "onesweepv":
        _, name = _run_and_compare(oracle, blob, offs, lay_t, n, P, rank, buf_size, algo, start, init, max_records,
                                   check_dest, capi.FLAG_OSV_ROUNDS_KERNEL)
        assert name in ("onesweepd", "generic"), name
    return first


def _run_and_compare(oracle, blob, offs, lay_t, n, P, rank, buf_size, algo, start, init, max_records, check_dest, flags):
    torch = _torch()
    from mimir_amd.packer import DevicePacker
    glay, olay = _layouts(lay_t)
    if init is not None:
        pytest.skip("init offsets are covered by test_append_across_calls")
    want = oracle.pack_round(olay, blob, offs, n, start, P, rank, buf_size)
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(np.ascontiguousarray(blob)).to(dev)
    d_off = None
    if offs is not None:
        assert int(offs[-1]) < (1 << 32)
        d_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).to(dev)
    pk = DevicePacker(glay, P, rank, buf_size, max_records or max(n, 1), flags=flags)
    image = torch.full((P * buf_size,), 0xEE, dtype=torch.uint8, device=dev)
    local = torch.full((max(len(blob), 16),), 0xDD, dtype=torch.uint8, device=dev) if rank >= 0 else None
    dest = torch.full((max(n - start, 1),), 0xFF, dtype=torch.uint8, device=dev)
    got = pk.pack(d_blob, n, image, start=start, rec_off=d_off, local_out=local,
                  dest_out=dest, algo=algo)
    torch.cuda.synchronize()
    assert got.next == want["next"], (got, want["next"])
    assert got.send_offset.tolist() == want["send_offset"].tolist()
    assert got.packed_records == want["packed_records"]
    assert got.local_records == want["local_records"]
    assert got.local_bytes == want["local_bytes"]
    img = image.cpu().numpy()
    for p in range(P):
        so = int(want["send_offset"][p])
        have = img[p * buf_size:p * buf_size + so]
        assert np.array_equal(have, want["partitions"][p]), f"partition {p} differs"
    if rank >= 0:
        assert np.array_equal(local.cpu().numpy()[:got.local_bytes], want["local"])
    if check_dest:
        _, d = oracle.dest(olay, blob, offs, n, P)
        consumed = got.next - start
        assert np.array_equal(dest.cpu().numpy()[:consumed].astype(np.int32),
                              d[start:got.next])
    algo_name = pk.last_algo()
    pk.close()
    return got, algo_name


# ---- 16-byte fixed records: the one-sweep kernel ---------------------------------

@pytest.mark.parametrize("P", [1, 2, 3, 4, 5, 7, 8, 13, 16, 33, 64])
def test_onesweep16_partitions_all_ranks(oracle, P):
    n = 20000 + P
    blob, _, lay = datagen.u64_records(n, seed=100 + P)
    got, name = run_and_compare(oracle, blob, None, lay, n, P, -1, 16 * n, capi.ALGO_ONESWEEP16)
    assert name == "onesweep16" and got.next == n


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 511, 512, 4095, 4096, 4097, 8192, 12289, 70001])
def test_onesweep16_ragged_sizes(oracle, n):
    blob, _, lay = datagen.u64_records(n, seed=n)
    run_and_compare(oracle, blob, None, lay, n, 8, -1, 16 * n, capi.ALGO_ONESWEEP16)
    run_and_compare(oracle, blob, None, lay, n, 8, 3, 16 * n, capi.ALGO_ONESWEEP16)


@pytest.mark.parametrize("rank", [0, 3, 7])
def test_onesweep16_local_bypass(oracle, rank):
    n = 50000
    blob, _, lay = datagen.u64_records(n, seed=5)
    got, _ = run_and_compare(oracle, blob, None, lay, n, 8, rank, 16 * n, capi.ALGO_AUTO)
    assert got.send_offset[rank] == 0 and got.local_records > 0


@pytest.mark.parametrize("buf_records", [1, 7, 100, 513, 4096, 5000])
def test_onesweep16_round_cut(oracle, buf_records):
    """A full partition ends the round at exactly the reference's record."""
    n = 60000
    blob, _, lay = datagen.u64_records(n, seed=77)
    for rank in (-1, 2):
        got, _ = run_and_compare(oracle, blob, None, lay, n, 8, rank, 16 * buf_records,
                                 capi.ALGO_ONESWEEP16)
        assert got.next < n


def test_onesweep16_rounds_chain_to_the_end(oracle):
    """Strict reference rounds: call after call, each starting at `next`."""
    torch = _torch()
    from mimir_amd.packer import DevicePacker
    n, P, rank, buf = 300000, 8, 5, 16 * 3000
    blob, _, lay_t = datagen.u64_records(n, seed=9)
    glay, olay = _layouts(lay_t)
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(blob).to(dev)
    pk = DevicePacker(glay, P, rank, buf, max_records=n)
    image = pk.new_send_image()
    local = torch.empty(len(blob), dtype=torch.uint8, device=dev)
    start, rounds = 0, 0
    while start < n:
        want = oracle.pack_round(olay, blob, None, n, start, P, rank, buf)
        got = pk.pack(d_blob, n, image, start=start, local_out=local)
        assert got.next == want["next"] > start
        assert got.send_offset.tolist() == want["send_offset"].tolist()
        img = image.cpu().numpy()
        for p in range(P):
            so = int(want["send_offset"][p])
            assert np.array_equal(img[p * buf:p * buf + so], want["partitions"][p])
        assert np.array_equal(local.cpu().numpy()[:got.local_bytes], want["local"])
        start = got.next
        rounds += 1
    assert rounds >= 10
    pk.close()


def test_append_across_calls(oracle):
    """Filling one round with several calls (init_offset) gives the same image
    as one call over the concatenated stream."""
    torch = _torch()
    from mimir_amd.packer import DevicePacker
    n, P, rank = 30000, 4, 1
    buf = 16 * n
    blob, _, lay_t = datagen.u64_records(n, seed=21)
    glay, olay = _layouts(lay_t)
    want = oracle.pack_round(olay, blob, None, n, 0, P, rank, buf)
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(blob).to(dev)
    pk = DevicePacker(glay, P, rank, buf, max_records=n)
    image = pk.new_send_image()
    local = torch.empty(len(blob), dtype=torch.uint8, device=dev)
    cuts = [0, 4097, 4098, 20000, n]
    init = np.zeros(P, dtype=np.int64)
    lbytes = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        for algo in (capi.ALGO_ONESWEEP16,):
            got = pk.pack(d_blob, b, image, start=a, init_offset=init,
                          local_out=local[lbytes:], algo=algo)
        assert got.next == b
        init = got.send_offset.copy()
        lbytes += got.local_bytes
    assert init.tolist() == want["send_offset"].tolist()
    img = image.cpu().numpy()
    for p in range(P):
        so = int(want["send_offset"][p])
        assert np.array_equal(img[p * buf:p * buf + so], want["partitions"][p])
    assert np.array_equal(local.cpu().numpy()[:lbytes], want["local"])
    pk.close()


def test_onesweep16_octant_keys_level8(oracle):
    """<char,uint64_t> with keycount 8: 8 key characters + u64 = 16 bytes."""
    n = 40000
    blob, _, lay = datagen.octree_records(n, level=8, seed=43)
    run_and_compare(oracle, blob, None, lay, n, 8, 6, 16 * n, capi.ALGO_AUTO)


@pytest.mark.parametrize("klen", [1, 3, 4, 5, 8, 11, 12, 13, 15, 16])
def test_onesweep16_key_widths(oracle, klen):
    """16-byte records whose key is the first klen bytes (value = the rest)."""
    n = 9000
    rng = np.random.default_rng(klen)
    blob = rng.integers(0, 256, size=16 * n, dtype=np.uint8)
    lay = (capi.KEY_FIXED, klen, capi.VAL_FIXED, 16 - klen) if klen < 16 else (
        capi.KEY_FIXED, 16, capi.VAL_VOID, 0)
    run_and_compare(oracle, blob, None, lay, n, 7, 2, 16 * n, capi.ALGO_ONESWEEP16)


def test_skewed_keys_one_hot_partition(oracle):
    """Every record to one destination: the hottest partition ends the round."""
    n = 20000
    rec = np.zeros((n, 2), dtype=np.uint64)
    rec[:, 0] = 12345
    rec[:, 1] = np.arange(n, dtype=np.uint64)
    blob = rec.view(np.uint8).reshape(-1)
    lay = (capi.KEY_FIXED, 8, capi.VAL_FIXED, 8)
    got, _ = run_and_compare(oracle, blob, None, lay, n, 8, -1, 16 * 5000, capi.ALGO_ONESWEEP16)
    assert got.next == 5000


# ---- generic three-pass path: every layout ---------------------------------------

@pytest.mark.parametrize("P,rank", [(1, -1), (2, 0), (4, 3), (7, -1), (8, 5), (64, 1), (200, 77)])
def test_generic_words(oracle, P, rank):
    n = 30000
    blob, offs, lay = datagen.word_records(n, n_unique=1000, alpha=0.5, len_mean=5, seed=44)
    got, name = run_and_compare(oracle, blob, offs, lay, n, P, rank, int(offs[-1]), capi.ALGO_GENERIC)
    assert name == "generic" and got.next == n


def test_generic_words_varied_lengths_and_cuts(oracle):
    n = 25000
    blob, offs, lay = datagen.word_records(n, n_unique=5000, alpha=1.0, len_mean=8, len_sd=2.5, seed=4)
    for buf in (64, 1000, 4096, 65536):
        got, _ = run_and_compare(oracle, blob, offs, lay, n, 8, 2, buf, capi.ALGO_AUTO)
        assert got.next < n
    # start in the middle of the stream
    run_and_compare(oracle, blob, offs, lay, n, 8, 2, 4096, capi.ALGO_AUTO, start=12345)


def test_generic_long_words(oracle):
    """Words up to 1023 characters (examples/wordcount.cpp:74)."""
    rng = np.random.default_rng(1)
    vocab = [bytes(rng.integers(97, 123, size=L, dtype=np.uint8)) for L in
             [1, 2, 3, 11, 12, 13, 23, 24, 25, 100, 500, 1023]]
    ranks = rng.integers(0, len(vocab), size=5000)
    blob, offs, lay = datagen.word_records_from(vocab, ranks)
    run_and_compare(oracle, blob, offs, lay, 5000, 8, 0, int(offs[-1]), capi.ALGO_AUTO)


@pytest.mark.parametrize("level", [1, 4, 7, 9, 12, 15])
def test_generic_octant_levels(oracle, level):
    """Odd fixed widths (level+8 bytes), unaligned records."""
    n = 20000
    blob, _, lay = datagen.octree_records(n, level=level, seed=level)
    run_and_compare(oracle, blob, None, lay, n, 8, 4, (level + 8) * n, capi.ALGO_AUTO)
    run_and_compare(oracle, blob, None, lay, n, 8, 4, (level + 8) * 333, capi.ALGO_AUTO)


def test_generic_join_records(oracle):
    n = 20000
    blob, offs, lay = datagen.join_records(n, n_unique=2000, alpha=1.0, len_mean=6, len_sd=1, seed=45)
    run_and_compare(oracle, blob, offs, lay, n, 4, 2, int(offs[-1]), capi.ALGO_AUTO)
    run_and_compare(oracle, blob, offs, lay, n, 8, -1, 8192, capi.ALGO_AUTO)


def test_generic_string_values(oracle):
    """<char*,char*>: both halves NUL-terminated."""
    rng = np.random.default_rng(3)
    recs, offs = [], [0]
    for _ in range(4000):
        k = bytes(rng.integers(97, 123, size=rng.integers(1, 20), dtype=np.uint8)) + b"\0"
        v = bytes(rng.integers(65, 91, size=rng.integers(0, 30), dtype=np.uint8)) + b"\0"
        recs.append(k + v)
        offs.append(offs[-1] + len(k) + len(v))
    blob = np.frombuffer(b"".join(recs), dtype=np.uint8)
    offs = np.array(offs, dtype=np.uint64)
    lay = (capi.KEY_CSTR, 0, capi.VAL_CSTR, 0)
    run_and_compare(oracle, blob, offs, lay, 4000, 5, 3, int(offs[-1]), capi.ALGO_AUTO)


def test_generic_equals_onesweep_on_16B(oracle):
    n = 33333
    blob, _, lay = datagen.u64_records(n, seed=8)
    for buf in (16 * n, 16 * 2000):
        a, na = run_and_compare(oracle, blob, None, lay, n, 8, 1, buf, capi.ALGO_GENERIC)
        b, nb = run_and_compare(oracle, blob, None, lay, n, 8, 1, buf, capi.ALGO_ONESWEEP16)
        assert (na, nb) == ("generic", "onesweep16") and a.next == b.next


# ---- byte-granular one-sweep kernel: variable width and short fixed widths -----------

@pytest.mark.parametrize("P,rank", [(1, -1), (1, 0), (2, 0), (3, 1), (4, 3), (7, -1), (8, 5), (8, -1),
                                    (16, 9), (33, 32), (63, 1), (64, -1)])
def test_onesweepv_words(oracle, P, rank):
    n = 30000
    blob, offs, lay = datagen.word_records(n, n_unique=1000, alpha=0.5, len_mean=5, seed=44)
    got, name = run_and_compare(oracle, blob, offs, lay, n, P, rank, int(offs[-1]), capi.ALGO_ONESWEEPV)
    assert name == "onesweepv" and got.next == n


def test_onesweepv_round_cuts_and_starts(oracle):
    n = 25000
    blob, offs, lay = datagen.word_records(n, n_unique=5000, alpha=1.0, len_mean=8, len_sd=2.5, seed=4)
    for buf in (64, 1000, 4096, 65536):
        got, name = run_and_compare(oracle, blob, offs, lay, n, 8, 2, buf, capi.ALGO_ONESWEEPV)
        assert got.next < n and name == "onesweepv"
    for start in (1, 1791, 1792, 12345, n - 1, n):
        _, name = run_and_compare(oracle, blob, offs, lay, n, 8, 2, 4096, capi.ALGO_ONESWEEPV, start=start)
    # a record larger than an empty partition is an error, as in the reference
    # (src/collectiveshuffler.h:156-159)
    torch = _torch()
    from mimir_amd.packer import DevicePacker
    pk = DevicePacker(_layouts(lay)[0], 8, 2, 8, n)
    d_blob = torch.from_numpy(np.ascontiguousarray(blob)).cuda()
    d_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).cuda()
    local = torch.empty(len(blob), dtype=torch.uint8, device="cuda:0")
    with pytest.raises(capi.MgsError) as ei:
        pk.pack(d_blob, n, pk.new_send_image(), rec_off=d_off, local_out=local, algo=capi.ALGO_ONESWEEPV)
    assert ei.value.status == capi.ERR_RECORD_SIZE
    pk.close()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 447, 448, 1791, 1792, 1793, 3584, 100000])
def test_onesweepv_ragged_sizes(oracle, n):
    blob, offs, lay = datagen.word_records(n, n_unique=300, alpha=0.8, len_mean=6, len_sd=2, seed=n)
    got, name = run_and_compare(oracle, blob, offs, lay, n, 8, 3, int(offs[-1]), capi.ALGO_ONESWEEPV)
    assert name == "onesweepv" and got.next == n


def test_onesweepv_unaligned_blob(oracle):
    """The record stream may start at any byte address."""
    torch = _torch()
    from mimir_amd.packer import DevicePacker
    n = 20000
    blob, offs, lay = datagen.word_records(n, n_unique=500, alpha=0.5, len_mean=5, len_sd=1.5, seed=9)
    glay, olay = _layouts(lay)
    want = oracle.pack_round(olay, blob, offs, n, 0, 8, -1, int(offs[-1]))
    for shift in (1, 3, 7, 15):
        buf = torch.zeros(len(blob) + 64, dtype=torch.uint8, device="cuda:0")
        view = buf[shift:shift + len(blob)]
        view.copy_(torch.from_numpy(np.ascontiguousarray(blob)))
        d_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).cuda()
        pk = DevicePacker(glay, 8, -1, int(offs[-1]), n)
        image = pk.new_send_image()
        got = pk.pack(view, n, image, rec_off=d_off, algo=capi.ALGO_ONESWEEPV)
        assert pk.last_algo() == "onesweepv"
        assert got.send_offset.tolist() == want["send_offset"].tolist()
        img = image.cpu().numpy()
        for p in range(8):
            so = int(want["send_offset"][p])
            assert np.array_equal(img[p * int(offs[-1]):p * int(offs[-1]) + so], want["partitions"][p])
        pk.close()


@pytest.mark.parametrize("level", [1, 4, 7, 9, 11, 12, 15])
def test_onesweepv_octant_levels(oracle, level):
    """Fixed widths other than 16 bytes, unaligned records (4, 2 or 1 records per lane)."""
    n = 20000
    blob, _, lay = datagen.octree_records(n, level=level, seed=level)
    _, name = run_and_compare(oracle, blob, None, lay, n, 8, 4, (level + 8) * n, capi.ALGO_ONESWEEPV)
    assert name == "onesweepv"
    run_and_compare(oracle, blob, None, lay, n, 8, 4, (level + 8) * 333, capi.ALGO_ONESWEEPV)


def test_onesweepv_string_values_and_join(oracle):
    rng = np.random.default_rng(3)
    recs, offs = [], [0]
    for _ in range(6000):
        k = bytes(rng.integers(97, 123, size=rng.integers(1, 8), dtype=np.uint8)) + b"\0"
        v = bytes(rng.integers(65, 91, size=rng.integers(0, 6), dtype=np.uint8)) + b"\0"
        recs.append(k + v)
        offs.append(offs[-1] + len(k) + len(v))
    blob = np.frombuffer(b"".join(recs), dtype=np.uint8)
    offs = np.array(offs, dtype=np.uint64)
    lay = (capi.KEY_CSTR, 0, capi.VAL_CSTR, 0)
    _, name = run_and_compare(oracle, blob, offs, lay, 6000, 5, 3, int(offs[-1]), capi.ALGO_ONESWEEPV)
    assert name == "onesweepv"
    blob, offs, lay = datagen.join_records(20000, n_unique=2000, alpha=1.0, len_mean=2, len_sd=0.5, seed=45)
    _, name = run_and_compare(oracle, blob, offs, lay, 20000, 4, 2, int(offs[-1]), capi.ALGO_ONESWEEPV)
    assert name == "onesweepv"


def test_onesweepv_adapts_tile_to_record_length(oracle):
    """The tile follows the record length: the first call on a handle looks at the ends of
    its offsets, later calls at what the call before saw.  A batch whose records are much
    longer than the last one's falls back to the generic kernels once, the next call uses
    a smaller tile.  Same bytes either way."""
    torch = _torch()
    from mimir_amd.packer import DevicePacker
    n = 30000
    blob, offs, lay = datagen.join_records(n, n_unique=2000, alpha=1.0, len_mean=30, len_sd=3, seed=46)
    glay, olay = _layouts(lay)
    total = int(offs[-1])
    want = oracle.pack_round(olay, blob, offs, n, 0, 8, 2, total)
    d_blob = torch.from_numpy(np.ascontiguousarray(blob)).cuda()
    d_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).cuda()
    pk = DevicePacker(glay, 8, 2, total, n)
    # short records first: the handle settles on its largest tile
    sblob, soffs, _ = datagen.join_records(n, n_unique=500, alpha=1.0, len_mean=2, len_sd=0, seed=47)
    simage = torch.empty(8 * total, dtype=torch.uint8, device="cuda:0")
    slocal = torch.empty(total, dtype=torch.uint8, device="cuda:0")
    pk.pack(torch.from_numpy(np.ascontiguousarray(sblob)).cuda(), n, simage,
            rec_off=torch.from_numpy(soffs.astype(np.uint32).view(np.int32)).cuda(), local_out=slocal)
    assert pk.last_algo() == "onesweepv"
    algos = []
    for _ in range(3):
        image = torch.full((8 * total,), 0xEE, dtype=torch.uint8, device="cuda:0")
        local = torch.full((total,), 0xDD, dtype=torch.uint8, device="cuda:0")
        got = pk.pack(d_blob, n, image, rec_off=d_off, local_out=local)
        algos.append(pk.last_algo())
        assert got.next == n and got.send_offset.tolist() == want["send_offset"].tolist()
        assert got.local_records == want["local_records"] and got.local_bytes == want["local_bytes"]
        img = image.cpu().numpy()
        for p in range(8):
            so = int(want["send_offset"][p])
            assert np.array_equal(img[p * total:p * total + so], want["partitions"][p])
        assert np.array_equal(local.cpu().numpy()[:got.local_bytes], want["local"])
    assert algos[0] == "generic" and algos[1] == "onesweepv" and algos[2] == "onesweepv", algos
    pk.close()


def test_short_string_records_take_the_quad_and_dword_kernel_until_long_ones_show(oracle):
    """Wordcount-sized records (16 bytes or less on average) run the register kernel that holds 20
    bytes of a record; records longer than that are copied the slow way and counted, and once more
    than one in 64 is that long the handle goes back to the staged kernel.  Same bytes either way."""
    torch = _torch()
    from mimir_amd.packer import DevicePacker
    n, P = 200_000, 8
    for share, expect_second in ((0.0, "k_onesweepr"), (0.06, "k_onesweepv")):
        rng = np.random.default_rng(int(share * 100) + 3)
        lens = np.where(rng.random(n) < share, rng.integers(13, 40, n), rng.integers(4, 8, n))
        words = [bytes(rng.integers(97, 123, size=int(l), dtype=np.uint8)) + b"\0" + int(i).to_bytes(8, "little")
                 for i, l in enumerate(lens)]
        blob = np.frombuffer(b"".join(words), dtype=np.uint8)
        offs = np.concatenate([[0], np.cumsum([len(w) for w in words])]).astype(np.uint64)
        lay = (capi.KEY_CSTR, 0, capi.VAL_FIXED, 8)
        glay, olay = _layouts(lay)
        total = int(offs[-1])
        want = oracle.pack_round(olay, blob, offs, n, 0, P, -1, total)
        d_blob = torch.from_numpy(np.ascontiguousarray(blob)).cuda()
        d_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).cuda()
        pk = DevicePacker(glay, P, -1, total, n)
        kernels = []
        for _ in range(2):
            image = torch.full((P * total,), 0xEE, dtype=torch.uint8, device="cuda:0")
            got = pk.pack(d_blob, n, image, rec_off=d_off)
            kernels.append(pk.last_kernel())
            assert got.next == n and got.send_offset.tolist() == want["send_offset"].tolist()
            img = image.cpu().numpy()
            for p in range(P):
                so = int(want["send_offset"][p])
                assert np.array_equal(img[p * total:p * total + so], want["partitions"][p]), (share, p)
        if os.environ.get("MGS_OSR") != "0" and not os.environ.get("MGS_OSR_FORCE"):
            assert kernels == ["k_onesweepr", expect_second], (share, kernels)
        pk.close()


def test_onesweepv_falls_back_on_long_records(oracle):
    """A tile whose records do not fit the LDS staging buffer: the call is served
    by the generic path, decided on the device."""
    rng = np.random.default_rng(1)
    vocab = [bytes(rng.integers(97, 123, size=L, dtype=np.uint8)) for L in
             [1, 2, 3, 11, 12, 13, 23, 24, 25, 100, 500, 1023]]
    ranks = rng.integers(0, len(vocab), size=5000)
    blob, offs, lay = datagen.word_records_from(vocab, ranks)
    _, name = run_and_compare(oracle, blob, offs, lay, 5000, 8, 0, int(offs[-1]), capi.ALGO_ONESWEEPV)
    assert name == "generic"
    _, name = run_and_compare(oracle, blob, offs, lay, 5000, 8, 0, 3000, capi.ALGO_AUTO)
    assert name == "generic"


def test_onesweepv_skewed_keys(oracle):
    """Every record to one destination, and a Zipf-heavy vocabulary."""
    n = 20000
    blob, offs, lay = datagen.word_records_from([b"same"], np.zeros(n, dtype=np.int64))
    run_and_compare(oracle, blob, offs, lay, n, 8, -1, int(offs[-1]), capi.ALGO_ONESWEEPV)
    run_and_compare(oracle, blob, offs, lay, n, 8, -1, 9000, capi.ALGO_ONESWEEPV)
    blob, offs, lay = datagen.word_records(n, n_unique=50, alpha=2.0, len_mean=4, seed=7)
    run_and_compare(oracle, blob, offs, lay, n, 8, 6, int(offs[-1]), capi.ALGO_ONESWEEPV)


def test_window_limit_max_records(oracle):
    """A call looks at max_records records at most."""
    n = 10000
    blob, _, lay = datagen.u64_records(n, seed=2)
    torch = _torch()
    from mimir_amd.packer import DevicePacker
    glay, olay = _layouts(lay)
    pk = DevicePacker(glay, 4, -1, 16 * n, max_records=4500)
    d_blob = torch.from_numpy(blob).cuda()
    image = pk.new_send_image()
    got = pk.pack(d_blob, n, image)
    assert got.next == 4500
    want = oracle.pack_round(olay, blob, None, 4500, 0, 4, -1, 16 * n)
    assert got.send_offset.tolist() == want["send_offset"].tolist()
    pk.close()


def test_errors_are_reported(oracle):
    torch = _torch()
    from mimir_amd.packer import DevicePacker
    # record larger than a partition: the reference aborts with "KV size ...
    # is larger than buf_size" (src/collectiveshuffler.h:156-159)
    blob, offs, lay = datagen.word_records(100, n_unique=5, len_mean=30, seed=1)
    glay, _ = _layouts(lay)
    pk = DevicePacker(glay, 2, -1, 16, max_records=100)
    d_blob = torch.from_numpy(blob).cuda()
    d_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).cuda()
    with pytest.raises(capi.MgsError) as e:
        pk.pack(d_blob, 100, pk.new_send_image(), rec_off=d_off)
    assert e.value.status == capi.ERR_RECORD_SIZE
    pk.close()
    # a key without its NUL is not a serialisation of <char*,uint64_t>
    bad = blob.copy()
    bad[int(offs[1]) - 9] = 65
    pk = DevicePacker(glay, 2, -1, 1 << 16, max_records=100)
    with pytest.raises(capi.MgsError) as e:
        pk.pack(torch.from_numpy(bad).cuda(), 100, pk.new_send_image(), rec_off=d_off)
    assert e.value.status == capi.ERR_MALFORMED
    pk.close()


def test_synth_generator_matches_host(oracle):
    torch = _torch()
    from mimir_amd.packer import synth_u64_records
    n = 10007
    dev = synth_u64_records(n, seed=42, first=5).cpu().numpy()
    host, _, _ = datagen.u64_records(n, seed=42, first=5)
    assert np.array_equal(dev, host)


def test_large_stream_properties():
    """Full-size properties that need no oracle: every record lands exactly
    once (per-partition checksum of checksums), each partition holds only its
    own destination, and values stay in emit order (stability)."""
    torch = _torch()
    from mimir_amd.packer import DevicePacker, synth_u64_records
    n, P = 1 << 24, 8
    rec = synth_u64_records(n, seed=42)
    glay = capi.Layout.make(capi.KEY_FIXED, 8, capi.VAL_FIXED, 8)
    buf = 16 * (n // P + n // 64)
    pk = DevicePacker(glay, P, -1, buf, max_records=n)
    image = pk.new_send_image()
    dest = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    got = pk.pack(rec, n, image, dest_out=dest)
    assert got.next == n and int(got.send_offset.sum()) == 16 * n
    words = rec.view(torch.int64).view(n, 2)
    total_key = int(words[:, 0].sum().item())  # wrap-around int64 sums
    total_val = int(words[:, 1].sum().item())
    ksum = vsum = 0
    counts = torch.bincount(dest.to(torch.int64), minlength=P).cpu().numpy()
    for p in range(P):
        cnt = int(got.send_offset[p]) // 16
        assert cnt == counts[p]
        part = image[p * buf:p * buf + 16 * cnt].view(torch.int64).view(cnt, 2)
        ksum += int(part[:, 0].sum().item())
        vsum += int(part[:, 1].sum().item())
        # stability: the value is the emit index, so it must increase strictly
        assert bool((part[1:, 1] > part[:-1, 1]).all().item())
        # membership: the source records with dest p, in order, are this partition
        idx = torch.nonzero(dest == p).squeeze(1)
        assert torch.equal(words[idx], part)
    wrap = lambda x: (x + (1 << 63)) % (1 << 64) - (1 << 63)
    assert wrap(ksum) == wrap(total_key) and wrap(vsum) == wrap(total_val)
    pk.close()


def test_full_size_properties(oracle):
    """The headline configuration at its full size (1e9 <u64,u64> records, 8
    partitions), checked through properties that need no full-size oracle run:
    every record lands exactly once (counts, wrap-around checksums), every
    partition holds exactly the records whose destination it is, in emit order
    (the value is the emit index), and the destinations of a random sample of
    records equal hashlittle(key) % 8 as the oracle computes it."""
    torch = _torch()
    from mimir_amd.packer import DevicePacker, synth_u64_records
    n, P = 1_000_000_000, 8
    rec = synth_u64_records(n, seed=42)
    glay = capi.Layout.make(capi.KEY_FIXED, 8, capi.VAL_FIXED, 8)
    buf = 16 * (n // P + n // 64 + 4096)
    pk = DevicePacker(glay, P, -1, buf, max_records=n)
    image = pk.new_send_image()
    dest = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    got = pk.pack(rec, n, image, dest_out=dest)
    assert pk.last_algo() == "onesweep16"
    assert got.next == n and int(got.send_offset.sum()) == 16 * n
    words = rec.view(torch.int64).view(n, 2)
    # a sample against the oracle
    rng = np.random.default_rng(1)
    pick = np.unique(np.concatenate([rng.integers(0, n, size=200000), [0, 1, n - 2, n - 1]]))
    sample = words[torch.from_numpy(pick).cuda()].cpu().numpy().view(np.uint8).reshape(-1)
    olay = O.Layout.make(O.KEY_FIXED, 8, O.VAL_FIXED, 8)
    _, want_dest = oracle.dest(olay, sample, None, len(pick), P)
    assert np.array_equal(dest[torch.from_numpy(pick).cuda()].cpu().numpy(), want_dest.astype(np.uint8))
    total_key = int(words[:, 0].sum().item())  # wrap-around int64 sums
    total_val = int(words[:, 1].sum().item())
    ksum = vsum = 0
    for p in range(P):
        cnt = int(got.send_offset[p]) // 16
        idx = torch.nonzero(dest == p).squeeze(1)
        assert idx.numel() == cnt
        part = image[p * buf:p * buf + 16 * cnt].view(torch.int64).view(cnt, 2)
        ksum += int(part[:, 0].sum().item())
        vsum += int(part[:, 1].sum().item())
        assert bool((part[1:, 1] > part[:-1, 1]).all().item())  # stable: emit order kept
        # exactly its own records: the value is the emit index, the key is that record's.
        # (Gathers go in chunks: one torch row gather of 1.25e8 rows returned 2^26 wrong
        # rows on this stack, while the same indices in chunks are right.)
        assert bool((part[:, 1] == idx).all().item())
        for s in range(0, cnt, 1 << 24):
            e = min(cnt, s + (1 << 24))
            assert bool((words[:, 0][idx[s:e]] == part[s:e, 0]).all().item())
        del idx
    wrap = lambda x: (x + (1 << 63)) % (1 << 64) - (1 << 63)
    assert wrap(ksum) == wrap(total_key) and wrap(vsum) == wrap(total_val)
    pk.close()


@pytest.mark.parametrize("algo", [capi.ALGO_GENERIC, capi.ALGO_ONESWEEPV])
@pytest.mark.parametrize("bad", ["embedded_nul", "embedded_nul_long_key", "zero_length", "no_nul", "trailing_byte"])
def test_malformed_records_are_refused_like_the_oracle(oracle, algo, bad):
    """A record that is not a serialisation of the layout: the oracle's key-length rule
    (strlen + 1 for a char* key, key and value are the whole record) refuses it, and so
    does every kernel path -- the same records, not only the well-formed ones, decide."""
    torch = _torch()
    from mimir_amd.packer import DevicePacker
    val = (1).to_bytes(8, "little")
    good = [b"alpha\0" + val, b"be\0" + val, b"gamma-delta-epsilon\0" + val] * 400
    bad_rec = {"embedded_nul": b"ab\0cd\0" + val,
               "embedded_nul_long_key": b"abcdefghijklmnop\0qrstuv\0" + val,
               "zero_length": b"",
               "no_nul": b"abcdef" + val,
               "trailing_byte": b"ab\0" + val + b"x"}[bad]
    for where in (0, 700, len(good)):
        recs = good[:where] + [bad_rec] + good[where:]
        offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint64)
        blob = np.frombuffer(b"".join(recs), dtype=np.uint8).copy()
        lay_t = (capi.KEY_CSTR, 0, capi.VAL_FIXED, 8)
        glay, olay = _layouts(lay_t)
        n = len(recs)
        with pytest.raises(ValueError):
            oracle.pack_round(olay, blob, offs, n, 0, 4, 1, 1 << 20)
        for flags in (0, capi.FLAG_OSV_ROUNDS_KERNEL):
            pk = DevicePacker(glay, 4, 1, 1 << 20, n, flags=flags)
            image = torch.empty(4 << 20, dtype=torch.uint8, device="cuda:0")
            local = torch.empty(1 << 20, dtype=torch.uint8, device="cuda:0")
            d_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).cuda()
            with pytest.raises(capi.MgsError) as ei:
                pk.pack(torch.from_numpy(blob).cuda(), n, image, rec_off=d_off, local_out=local, algo=algo)
            assert ei.value.status == capi.ERR_MALFORMED, (bad, where, pk.last_algo())
            pk.close()


def test_two_string_keys_with_fixed_value(oracle):
    """keycount = 2 of char* with a fixed-width value: the key holds two NULs
    (bytestream<char*>::size sums strlen + 1 over the strings, src/serializer.h:193-198)."""
    rng = np.random.default_rng(3)
    recs = []
    for i in range(5000):
        a = bytes(rng.integers(97, 123, size=rng.integers(1, 6), dtype=np.uint8))
        b = bytes(rng.integers(97, 123, size=rng.integers(0, 5), dtype=np.uint8))
        recs.append(a + b"\0" + b + b"\0" + int(i).to_bytes(8, "little"))
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint64)
    blob = np.frombuffer(b"".join(recs), dtype=np.uint8).copy()
    torch = _torch()
    from mimir_amd.packer import DevicePacker
    glay = capi.Layout.make(capi.KEY_CSTR, 0, capi.VAL_FIXED, 8, key_count=2)
    olay = O.Layout.make(capi.KEY_CSTR, 0, capi.VAL_FIXED, 8, key_count=2)
    n, P, rank, buf = len(recs), 4, 2, 20000
    want = oracle.pack_round(olay, blob, offs, n, 0, P, rank, buf)
    for algo in (capi.ALGO_GENERIC, capi.ALGO_ONESWEEPV):
        pk = DevicePacker(glay, P, rank, buf, n)
        image = torch.empty(P * buf, dtype=torch.uint8, device="cuda:0")
        local = torch.empty(len(blob), dtype=torch.uint8, device="cuda:0")
        d_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).cuda()
        got = pk.pack(torch.from_numpy(blob).cuda(), n, image, rec_off=d_off, local_out=local, algo=algo)
        assert got.next == want["next"] and got.send_offset.tolist() == want["send_offset"].tolist()
        img = image.cpu().numpy()
        for p in range(P):
            so = int(want["send_offset"][p])
            assert np.array_equal(img[p * buf:p * buf + so], want["partitions"][p])
        pk.close()


def test_pack_vectors_boundary_sizes_on_device():
    """tests/golden/pack_vectors.json (the reference's own Serializer::kv_to_bytes and
    get_hash_code): a record goes into a partition with `bufsize` bytes left iff the
    reference's call returned its length rather than -1 (exact fit succeeds, one byte less
    ends the round, src/collectiveshuffler.h:143-162), it lands as the reference wrote it,
    and its destination is the reference's hash code modulo the rank count."""
    import json
    import os
    torch = _torch()
    from mimir_amd.packer import DevicePacker
    with open(os.path.join(os.path.dirname(__file__), "golden", "pack_vectors.json")) as f:
        vectors = json.load(f)["vectors"]
    dev = torch.device("cuda:0")
    algos = set()
    for v in vectors:
        if v["shape"] == "cstr_u64":
            glay = capi.Layout.make(capi.KEY_CSTR, 0, capi.VAL_FIXED, 8, key_count=v["keycount"])
        elif v["shape"] == "u64_u64":
            glay = capi.Layout.make(capi.KEY_FIXED, 8, capi.VAL_FIXED, 8)
        elif v["shape"] == "chars_u64":
            glay = capi.Layout.make(capi.KEY_FIXED, v["keycount"], capi.VAL_FIXED, 8)
        else:
            glay = capi.Layout.make(capi.KEY_CSTR, 0, capi.VAL_FIXED, 16)
        rec = np.frombuffer(bytes.fromhex(v["record_hex"]), dtype=np.uint8)
        R = len(rec)
        d_rec = torch.from_numpy(rec.copy()).to(dev)
        d_off = None if glay.fixed_len() else torch.tensor([0, R], dtype=torch.int32, device=dev)
        # (16-byte records take the 16-byte kernel only with 16-byte aligned fills: both paths are seen)
        for P in (1, 7):
            want_d = v["hash_code"] % P
            buf = 64
            pk = DevicePacker(glay, P, -1, buf, max_records=1)
            for f in v["fits"]:
                init = np.full(P, buf - f["bufsize"], dtype=np.int64)
                image = torch.full((P * buf,), 0xEE, dtype=torch.uint8, device=dev)
                dest = torch.full((1,), 0xFF, dtype=torch.uint8, device=dev)
                got = pk.pack(d_rec, 1, image, rec_off=d_off, init_offset=init, dest_out=dest)
                algos.add(pk.last_kernel())
                assert (got.next == 1) == (f["ret"] != -1), (v["shape"], v["key"], f, pk.last_kernel())
                if f["ret"] != -1:
                    at = want_d * buf + buf - f["bufsize"]
                    img = image.cpu().numpy()
                    assert img[at:at + R].tobytes() == rec.tobytes(), (v["shape"], v["key"], f)
                    assert int(got.send_offset[want_d]) == buf - f["bufsize"] + R
                    assert int(dest.cpu()[0]) == want_d
            pk.close()
        # a partition of exactly the record's size takes it; one byte smaller can never take it
        pk = DevicePacker(glay, 1, -1, R, max_records=1)
        image = torch.full((R,), 0xEE, dtype=torch.uint8, device=dev)
        assert pk.pack(d_rec, 1, image, rec_off=d_off).next == 1
        assert image.cpu().numpy().tobytes() == rec.tobytes()
        pk.close()
        if R > 1:
            pk = DevicePacker(glay, 1, -1, R - 1, max_records=1)
            image = torch.full((R,), 0xEE, dtype=torch.uint8, device=dev)
            with pytest.raises(capi.MgsError) as ei:
                pk.pack(d_rec, 1, image, rec_off=d_off)
            assert ei.value.status == capi.ERR_RECORD_SIZE
            pk.close()
    assert {"k_onesweepv", "k_onesweepr"} <= algos, algos
