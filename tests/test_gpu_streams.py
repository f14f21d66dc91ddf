"""Streams layout (mgs_pack_streams_device) against the oracle driven round by round:
every round's partition -- a view [round_base, round_base + send_offset) into a
destination stream -- byte-identical to the oracle's send buffer of that round, the
same round boundaries, local stream and per-round counters as the reference's
sequence of exchange_kv rounds (src/collectiveshuffler.h:131-166), and the same
result as the image layout (mgs_pack_rounds_device)."""
import numpy as np
import pytest

from mimir_amd import capi, datagen
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def oracle_rounds(oracle, olay, blob, offs, n, P, rank, buf, start=0, init=None, max_rounds=1 << 30):
    """The reference's rounds: pack until a record does not fit, exchange, go on."""
    rounds, cur = [], start
    while len(rounds) < max_rounds:
        r = oracle.pack_round(olay, blob, offs, n, cur, P, rank, buf,
                              init_offset=init if not rounds else None)
        rounds.append(r)
        cur = r["next"]
        if cur >= n:
            break
    return rounds, cur


def check_streams(oracle, blob, offs, lay_t, n, P, rank, buf, max_rounds, start=0, fill=None, base=None,
                  stream_cap=None, expect_algo=None, prefilled=None):
    """fill / base: bytes already in the streams and where the open round began (the oracle
    sees init_offset = fill - base); prefilled: what those bytes are."""
    import torch
    from mimir_amd.packer import DevicePacker
    glay, olay = capi.Layout.make(*lay_t), O.Layout.make(*lay_t)
    init = None
    if fill is not None:
        init = np.asarray(fill, dtype=np.int64) - (np.asarray(base, dtype=np.int64) if base is not None else fill)
    want, want_next = oracle_rounds(oracle, olay, blob, offs, n, P, rank, buf, start, init, max_rounds)
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(np.ascontiguousarray(blob)).to(dev)
    d_off = None if offs is None else torch.from_numpy(offs.astype(np.uint32).view(np.int32)).to(dev)
    if stream_cap is None:
        stream_cap = (len(blob) + (int(max(fill)) if fill is not None else 0) + 4096) // 16 * 16
    pk = DevicePacker(glay, P, rank, buf, max_records=max(n, 1))
    streams = torch.full((P * stream_cap,), 0xEE, dtype=torch.uint8, device=dev)
    local = torch.full((max(len(blob), 16),), 0xDD, dtype=torch.uint8, device=dev) if rank >= 0 else None
    dest = torch.full((max(n - start, 1),), 0xFF, dtype=torch.uint8, device=dev)
    got = pk.pack_streams(d_blob, n, streams, stream_cap, max_rounds, start=start, rec_off=d_off,
                          stream_fill=fill, open_round_base=base, local_out=local, dest_out=dest)
    torch.cuda.synchronize()
    if expect_algo:
        assert pk.last_algo() == expect_algo
    assert got.n_rounds == len(want), (got.n_rounds, len(want))
    assert got.next == want_next
    img = streams.cpu().numpy()
    loc = local.cpu().numpy() if rank >= 0 else None
    lpos = 0
    for k, w in enumerate(want):
        assert int(got.round_next[k]) == w["next"], f"round {k}"
        assert got.round_send_offset[k].tolist() == w["send_offset"].tolist(), f"round {k}"
        assert int(got.round_local_records[k]) == w["local_records"], f"round {k}"
        assert int(got.round_local_bytes[k]) == w["local_bytes"], f"round {k}"
        for p in range(P):
            if p == rank:
                continue
            so = int(w["send_offset"][p])
            o = p * stream_cap + int(got.round_base[k][p])
            lo = int(init[p]) if (init is not None and k == 0) else 0   # bytes that were there before the call
            assert np.array_equal(img[o + lo:o + so], w["partitions"][p][lo:]), f"round {k} partition {p}"
            if k + 1 < len(want):  # the rounds of a stream lie back to back
                assert int(got.round_base[k + 1][p]) == int(got.round_base[k][p]) + so
        if rank >= 0:
            assert np.array_equal(loc[lpos:lpos + w["local_bytes"]], w["local"]), f"round {k} local stream"
            lpos += w["local_bytes"]
    for p in range(P):
        if p != rank:
            first = int(base[p]) if base is not None else (int(fill[p]) if fill is not None else 0)
            assert int(got.round_base[0][p]) == first
    assert got.local_bytes == lpos
    assert got.packed_records == sum(w["packed_records"] for w in want)
    assert got.local_records == sum(w["local_records"] for w in want)
    _, d = oracle.dest(olay, blob, offs, n, P)
    assert np.array_equal(dest.cpu().numpy()[:want_next - start].astype(np.int32), d[start:want_next])
    pk.close()
    return got


@pytest.mark.parametrize("P,rank", [(8, -1), (8, 3), (2, 0), (3, 2), (5, -1), (64, 17), (1, -1)])
def test_streams_16_byte_records(oracle, P, rank):
    n = 200_000
    blob, offs, lay = datagen.u64_records(n, seed=100 + P)
    buf = 16 * (n // P // 9)          # about ten rounds
    got = check_streams(oracle, blob, offs, lay, n, P, rank, buf, max_rounds=64, expect_algo="onesweep16")
    assert got.n_rounds >= (2 if P == 1 else 8)


@pytest.mark.parametrize("cap_records", [1, 2, 7, 63, 64, 65, 500, 3583, 3584, 3585])
def test_streams_tiny_and_tile_sized_partitions(oracle, cap_records):
    """Many cuts inside one tile (3584 records), cuts on tile and row edges."""
    n = 9000 if cap_records < 100 else 40_000
    blob, offs, lay = datagen.u64_records(n, seed=7)
    max_rounds = 4096 if cap_records < 8 else 700
    check_streams(oracle, blob, offs, lay, n, 4, 1, 16 * cap_records, max_rounds=max_rounds)


def test_streams_many_tiles_per_round(oracle):
    """Rounds hundreds of tiles long: the finder's probes gallop, bracket and refine."""
    n, P = 3_000_000, 8
    blob, offs, lay = datagen.u64_records(n, seed=13)
    got = check_streams(oracle, blob, offs, lay, n, P, -1, 16 * 60_000, max_rounds=16)
    assert 6 <= got.n_rounds <= 8
    got = check_streams(oracle, blob, offs, lay, n, P, 6, 16 * 400_000, max_rounds=4)   # no cut at all
    assert got.n_rounds == 1


def test_streams_rounds_run_out_and_go_on(oracle):
    """max_rounds smaller than the rounds of the batch: the call stops at a cut; the next
    call goes on from there in the same streams."""
    n, P, rank = 150_000, 8, 5
    blob, offs, lay = datagen.u64_records(n, seed=3)
    buf = 16 * (n // P // 20)
    got = check_streams(oracle, blob, offs, lay, n, P, rank, buf, max_rounds=6)
    assert got.n_rounds == 6 and got.next < n
    got2 = check_streams(oracle, blob, offs, lay, n, P, rank, buf, max_rounds=64, start=got.next)
    assert got2.next == n


def test_streams_with_fill_and_open_round(oracle):
    """Streams that already hold bytes, the open round among them: the first cut comes
    when fill - base + new bytes pass buf_size."""
    n, P, rank = 100_000, 4, -1
    blob, offs, lay = datagen.u64_records(n, seed=11)
    buf = 16 * 3000
    fill = np.array([16 * 5000, 16 * 40, 16 * 3000, 16 * 17], dtype=np.int64)
    base = np.array([16 * 2001, 16 * 40, 0, 0], dtype=np.int64)   # open round: 2999, 0, 3000 (full), 17 records
    check_streams(oracle, blob, offs, lay, n, P, rank, buf, max_rounds=40, fill=fill, base=base)
    check_streams(oracle, blob, offs, lay, n, P, rank, buf, max_rounds=40, fill=fill)   # open round empty


def test_streams_full_stream_ends_the_call(oracle):
    """A stream that runs full: the call ends there, the last round is open and ends with
    the last record that was packed; rounds before it are complete."""
    import torch
    from mimir_amd.packer import DevicePacker
    n, P = 100_000, 4
    blob, offs, lay = datagen.u64_records(n, seed=17)
    buf, cap = 16 * 2000, 16 * 9000
    glay, olay = capi.Layout.make(*lay), O.Layout.make(*lay)
    # where the first stream fills up: one round of the oracle with partitions as large as the streams
    clip = oracle.pack_round(olay, blob, offs, n, 0, P, -1, cap)
    assert clip["next"] < n
    want, _ = oracle_rounds(oracle, olay, blob, offs, clip["next"], P, -1, buf)   # the reference on the packed records
    pk = DevicePacker(glay, P, -1, buf, max_records=n)
    streams = torch.full((P * cap,), 0xEE, dtype=torch.uint8, device="cuda:0")
    got = pk.pack_streams(torch.from_numpy(blob).cuda(), n, streams, cap, 64)
    assert got.next == clip["next"] and got.n_rounds == len(want)
    img = streams.cpu().numpy()
    for k, w in enumerate(want):
        assert int(got.round_next[k]) == w["next"]
        assert got.round_send_offset[k].tolist() == w["send_offset"].tolist()
        for p in range(P):
            o, so = p * cap + int(got.round_base[k][p]), int(w["send_offset"][p])
            assert np.array_equal(img[o:o + so], w["partitions"][p])
    pk.close()


def test_streams_skewed_keys(oracle):
    """One hot destination: rounds end while the other partitions are almost empty."""
    n, P = 120_000, 8
    rec = np.zeros((n, 2), dtype=np.uint64)
    rng = np.random.default_rng(5)
    hot = rng.random(n) < 0.8
    rec[:, 0] = np.where(hot, 12345, rng.integers(0, 1 << 62, size=n, dtype=np.uint64))
    rec[:, 1] = np.arange(n)
    blob = rec.view(np.uint8).reshape(-1)
    check_streams(oracle, blob, None, (capi.KEY_FIXED, 8, capi.VAL_FIXED, 8), n, P, 2, 16 * 4000, max_rounds=64)


@pytest.mark.parametrize("kind", ["words", "oct12", "join", "oct15"])
@pytest.mark.parametrize("rank", [1, -1])
def test_streams_other_layouts(oracle, kind, rank):
    n = 60_000
    if kind == "words":
        blob, offs, lay = datagen.word_records(n, n_unique=3000, alpha=1.0, len_mean=6, len_sd=1, seed=21)
    elif kind == "oct12":
        blob, offs, lay = datagen.octree_records(n, level=12, seed=22)
    elif kind == "oct15":
        blob, offs, lay = datagen.octree_records(n, level=15, seed=24)
    else:
        blob, offs, lay = datagen.join_records(n, n_unique=2000, len_mean=6, len_sd=1, seed=23)
    P = 4
    buf = (len(blob) // P // 7) // 16 * 16
    got = check_streams(oracle, blob, offs, lay, n, P, rank, buf, max_rounds=64, expect_algo="onesweepv")
    assert got.n_rounds >= 5


@pytest.mark.parametrize("buf", [40, 64, 301, 5000])
def test_streams_words_tiny_partitions(oracle, buf):
    """Byte-granular records, partitions of a few records: many cuts per tile, byte-exact limits."""
    n = 6000
    blob, offs, lay = datagen.word_records(n, n_unique=300, alpha=0.7, len_mean=7, len_sd=2, seed=31)
    check_streams(oracle, blob, offs, lay, n, 5, 2, buf, max_rounds=4096)


def test_streams_equal_images(oracle):
    """The two layouts report the same rounds; round k's partitions hold the same bytes."""
    import torch
    from mimir_amd.packer import DevicePacker
    n, P, rank = 400_000, 8, 4
    blob, _, lay = datagen.u64_records(n, seed=9)
    buf = 16 * (n // P // 12)
    pk = DevicePacker(capi.Layout.make(*lay), P, rank, buf, max_records=n)
    d_blob = torch.from_numpy(blob).cuda()
    images = torch.empty(40 * P * buf, dtype=torch.uint8, device="cuda:0")
    local_a = torch.empty(len(blob), dtype=torch.uint8, device="cuda:0")
    local_b = torch.empty(len(blob), dtype=torch.uint8, device="cuda:0")
    a = pk.pack_rounds(d_blob, n, images, P * buf, 40, local_out=local_a)
    cap = 16 * n
    streams = torch.empty(P * cap, dtype=torch.uint8, device="cuda:0")
    pk.pack_streams_async(d_blob, n, streams, cap, 40, local_out=local_b)
    b = pk.streams_result()
    assert a.n_rounds == b.n_rounds and a.next == b.next == n
    assert np.array_equal(a.round_send_offset, b.round_send_offset)
    assert np.array_equal(a.round_next, b.round_next)
    assert np.array_equal(a.round_local_bytes, b.round_local_bytes)
    assert torch.equal(local_a[:a.local_bytes], local_b[:b.local_bytes])
    for k in range(a.n_rounds):
        for p in range(P):
            so = int(a.round_send_offset[k][p])
            oa, ob = k * P * buf + p * buf, p * cap + int(b.round_base[k][p])
            assert torch.equal(images[oa:oa + so], streams[ob:ob + so])
    pk.close()


def test_streams_record_larger_than_partition(oracle):
    import torch
    from mimir_amd.packer import DevicePacker
    n = 5000
    blob, offs, lay = datagen.word_records(n, n_unique=100, alpha=0.5, len_mean=9, len_sd=3, seed=41)
    pk = DevicePacker(capi.Layout.make(*lay), 4, -1, 12, max_records=n)    # shorter than most records
    streams = torch.empty(4 * 2 * len(blob), dtype=torch.uint8, device="cuda:0")
    d_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).cuda()
    with pytest.raises(capi.MgsError) as e:
        pk.pack_streams(torch.from_numpy(blob).cuda(), n, streams, 2 * len(blob), 4096, rec_off=d_off)
    assert e.value.status == capi.ERR_RECORD_SIZE
    pk.close()


@pytest.mark.parametrize("kind", ["u64", "words"])
def test_streams_chained_calls(oracle, kind):
    """A stream of records handed over in three calls that go on in the same destination
    streams (stream_fill = where the last call stopped, open_round_base = where its open
    round began): the rounds, taken together, are those of one call over everything."""
    import torch
    from mimir_amd.packer import DevicePacker
    n, P, rank = 90_000, 4, 2
    if kind == "u64":
        blob, offs, lay = datagen.u64_records(n, seed=51)
        buf = 16 * 2500
    else:
        blob, offs, lay = datagen.word_records(n, n_unique=2000, alpha=1.0, len_mean=6, len_sd=1, seed=52)
        buf = 40_000
    glay, olay = capi.Layout.make(*lay), O.Layout.make(*lay)
    want, _ = oracle_rounds(oracle, olay, blob, offs, n, P, rank, buf)
    cap = (len(blob) + 4096) // 16 * 16
    pk = DevicePacker(glay, P, rank, buf, max_records=n)
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(blob).to(dev)
    d_off = None if offs is None else torch.from_numpy(offs.astype(np.uint32).view(np.int32)).to(dev)
    streams = torch.full((P * cap,), 0xEE, dtype=torch.uint8, device=dev)
    local = torch.empty(len(blob), dtype=torch.uint8, device=dev)
    fill = base = None
    rounds, lused = [], 0   # (base row, send_offset row, next, local records) of every round, merged over the calls
    for n_end in (n // 3 + 5, 2 * n // 3 + 1, n):
        start = rounds[-1][2] if rounds else 0
        got = pk.pack_streams(d_blob, n_end, streams, cap, 64, start=start, rec_off=d_off, stream_fill=fill,
                              open_round_base=base, local_out=local[lused:])
        assert got.next == n_end
        for k in range(got.n_rounds):
            row = (got.round_base[k].copy(), got.round_send_offset[k].copy(), int(got.round_next[k]),
                   int(got.round_local_records[k]))
            if k == 0 and rounds:   # the open round of the call before goes on
                prev = rounds.pop()
                assert np.array_equal(prev[0], row[0])
                row = (row[0], row[1], row[2], prev[3] + row[3])
            rounds.append(row)
        lused += got.local_bytes
        base = got.round_base[got.n_rounds - 1].copy()
        fill = base + got.round_send_offset[got.n_rounds - 1]
        fill[rank] = 0
    assert len(rounds) == len(want)
    img = streams.cpu().numpy()
    for k, w in enumerate(want):
        b, so, nxt, lrec = rounds[k]
        assert nxt == w["next"] and so.tolist() == w["send_offset"].tolist() and lrec == w["local_records"], f"round {k}"
        for p in range(P):
            if p != rank:
                o = p * cap + int(b[p])
                assert np.array_equal(img[o:o + int(so[p])], w["partitions"][p]), f"round {k} partition {p}"
    assert np.array_equal(local.cpu().numpy()[:lused], np.concatenate([w["local"] for w in want]))
    pk.close()


@pytest.mark.parametrize("rank", [1, -1])
def test_streams_records_too_long_for_the_one_sweep_kernels(oracle, rank):
    """Words of a hundred bytes and more: a tile of them exceeds the one-sweep kernels' LDS
    image, so the call packs round by round with the generic kernels and appends every
    round's partitions to the streams -- same rounds, same bytes, same views; a second call
    on the same handle goes on in the open round without trying the one-sweep kernel again."""
    import torch
    from mimir_amd.packer import DevicePacker
    n, P, buf = 6000, 3, 16384
    blob, offs, lay = datagen.word_records(n, n_unique=300, alpha=0.7, len_mean=110, len_sd=30, seed=5)
    got = check_streams(oracle, blob, offs, lay, n, P, rank, buf, 256, expect_algo="generic")
    assert got.n_rounds >= 5
    # chained: the second half goes on where the first stopped
    glay, olay = capi.Layout.make(*lay), O.Layout.make(*lay)
    want, _ = oracle_rounds(oracle, olay, blob, offs, n, P, rank, buf)
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(np.ascontiguousarray(blob)).to(dev)
    d_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).to(dev)
    cap = (len(blob) + 4096) // 16 * 16
    pk = DevicePacker(glay, P, rank, buf, max_records=n)
    streams = torch.zeros(P * cap, dtype=torch.uint8, device=dev)
    local = torch.zeros(len(blob), dtype=torch.uint8, device=dev)
    a = pk.pack_streams(d_blob, n // 2, streams, cap, 256, rec_off=d_off, local_out=local)
    assert pk.last_algo() == "generic" and a.next == n // 2
    base = a.round_base[-1].copy()
    fill = base + a.round_send_offset[-1]
    b = pk.pack_streams(d_blob, n, streams, cap, 256, start=n // 2, rec_off=d_off, stream_fill=fill,
                        open_round_base=base, local_out=local[a.local_bytes:])
    assert b.next == n and a.n_rounds + b.n_rounds - 1 == len(want)
    img = streams.cpu().numpy()
    so = np.concatenate([a.round_send_offset[:-1], b.round_send_offset])
    rb = np.concatenate([a.round_base[:-1], b.round_base])
    for k, w in enumerate(want):
        assert so[k].tolist() == w["send_offset"].tolist(), f"round {k}"
        for p in range(P):
            if p != rank:
                o = p * cap + int(rb[k][p])
                assert np.array_equal(img[o:o + int(so[k][p])], w["partitions"][p]), f"round {k} partition {p}"
    # a stream without room for a whole partition: the call stops before that round
    small = (int(so[:2].sum(axis=0).max()) + buf // 2) // 16 * 16
    streams2 = torch.zeros(P * small, dtype=torch.uint8, device=dev)
    c = pk.pack_streams(d_blob, n, streams2, small, 256, rec_off=d_off, local_out=local)
    assert 0 < c.next < n and c.n_rounds < 256
    assert c.next == want[c.n_rounds - 2]["next"]   # the rounds before the open (empty) one are the reference's
    assert c.round_send_offset[c.n_rounds - 1].sum() == 0
    pk.close()


def test_streams_empty_call_and_wrong_result_call(oracle):
    import torch
    from mimir_amd.packer import DevicePacker
    n, P = 1000, 4
    blob, _, lay = datagen.u64_records(n, seed=61)
    pk = DevicePacker(capi.Layout.make(*lay), P, 1, 16 * 100, max_records=n)
    streams = torch.empty(P * 16 * n, dtype=torch.uint8, device="cuda:0")
    local = torch.empty(16 * n, dtype=torch.uint8, device="cuda:0")
    d = torch.from_numpy(blob).cuda()
    fill = np.array([160, 0, 320, 16], dtype=np.int64)
    base = np.array([0, 0, 160, 16], dtype=np.int64)
    r = pk.pack_streams(d, 500, streams, 16 * n, 8, start=500, stream_fill=fill, open_round_base=base, local_out=local)
    assert r.n_rounds == 1 and r.next == 500 and r.packed_records == 0
    assert r.round_send_offset[0].tolist() == [160, 0, 160, 0] and r.round_base[0].tolist() == [0, 0, 160, 16]
    # a streams call in flight answers only to the streams result call
    pk.pack_streams_async(d, n, streams, 16 * n, 64, local_out=local)
    with pytest.raises(capi.MgsError) as e:
        pk.rounds_result()
    assert e.value.status == capi.ERR_STATE
    got = pk.streams_result()
    assert got.next == n and got.n_rounds > 1
    # ... and a rounds call asked for its result through the one-round call stays in flight:
    # the rounds result call still fetches it
    images = torch.empty(64 * P * 1600, dtype=torch.uint8, device="cuda:0")
    pk.pack_rounds_async(d, n, images, P * 1600, 64, local_out=local)
    with pytest.raises(capi.MgsError) as e:
        pk.result()
    assert e.value.status == capi.ERR_STATE
    got2 = pk.rounds_result()
    assert got2.next == n and got2.n_rounds == got.n_rounds
    assert got2.round_send_offset.tolist() == got.round_send_offset.tolist()
    pk.close()


def test_full_size_streams_at_the_reference_geometry(oracle):
    """The benchmarked call at the benchmarked size: 1e9 seed-42 <u64,u64> records, P = 8,
    MIMIR_COMM_SIZE = 64M (8 MiB partitions), streams layout.  Checked on the device:
    the streams hold what the one-round image holds, the rounds lie back to back, every
    round ends exactly where the next record of some destination no longer fits
    (src/collectiveshuffler.h:143-162), and four rounds -- first, second, middle, last --
    are byte-equal to the oracle driven from the previous round's end on a host copy of
    just their records."""
    import torch
    from mimir_amd.packer import DevicePacker, synth_u64_records
    n, P = 1_000_000_000, 8
    buf, _, tlb = capi.geometry(64 << 20, P)
    assert buf == 8 << 20 and tlb == 0
    dev = torch.device("cuda:0")
    rec = synth_u64_records(n, seed=42)
    glay = capi.Layout.make(capi.KEY_FIXED, 8, capi.VAL_FIXED, 8)
    olay = O.Layout.make(O.KEY_FIXED, 8, O.VAL_FIXED, 8)
    total = 16 * n
    max_rounds = int(total / P / buf * 1.15) + 8
    stream_cap = (total // P + total // 64 + 65536) // 16 * 16
    pk = DevicePacker(glay, P, -1, buf, max_records=n)
    streams = torch.empty(P * stream_cap, dtype=torch.uint8, device=dev)
    got = pk.pack_streams(rec, n, streams, stream_cap, max_rounds)
    assert pk.last_algo() == "onesweep16"
    # (iv) the round count of this batch, and nothing lost
    assert got.next == n and got.n_rounds == 239
    so, base, nxt = got.round_send_offset, got.round_base, got.round_next.astype(np.int64)
    assert int(so.sum()) == total
    # (ii) rounds lie back to back in every stream, the first at byte 0
    assert not base[0].any()
    assert np.array_equal(base[1:], base[:-1] + so[:-1])
    # (iii) no partition above its capacity; the record that opened round k + 1 did not fit round k
    assert int(so.max()) <= buf and int(so.min()) >= 0
    assert np.all(np.diff(nxt) > 0) and nxt[-1] == n
    fill = base[-1] + so[-1]

    # (i) against the one-round image of the same records (oracle-sampled in test_full_size_properties)
    big = 16 * (n // P + n // 64 + 4096)
    pk1 = DevicePacker(glay, P, -1, big, max_records=n)
    image = pk1.new_send_image()
    dest = torch.empty(n, dtype=torch.uint8, device=dev)
    one = pk1.pack(rec, n, image, dest_out=dest)
    assert one.next == n and one.send_offset.tolist() == fill.tolist()
    for p in range(P):
        f = int(fill[p])
        assert torch.equal(streams[p * stream_cap:p * stream_cap + f], image[p * big:p * big + f]), f"stream {p}"
    del image
    openers = torch.from_numpy(nxt[:-1]).to(dev)
    d_open = dest[openers].cpu().numpy().astype(np.int64)
    k = np.arange(got.n_rounds - 1)
    assert np.all(so[k, d_open] + 16 > buf), "a round ended although the next record fitted"
    # sampled destinations against the oracle
    rng = np.random.default_rng(2)
    pick = np.unique(np.concatenate([rng.integers(0, n, size=100000), nxt[:-1], nxt[:-1] - 1, [0, n - 1]]))
    words = rec.view(torch.int64).view(n, 2)
    sample = words[torch.from_numpy(pick).to(dev)].cpu().numpy().view(np.uint8).reshape(-1)
    _, want_dest = oracle.dest(olay, sample, None, len(pick), P)
    assert np.array_equal(dest[torch.from_numpy(pick).to(dev)].cpu().numpy(), want_dest.astype(np.uint8))
    del dest

    # (v) four rounds against the oracle, each driven from the end of the round before
    for kk in (0, 1, got.n_rounds // 2, got.n_rounds - 1):
        s = int(nxt[kk - 1]) if kk else 0
        e = min(n, int(nxt[kk]) + 4096)            # a few records beyond the cut, so that the oracle finds it
        host = rec[16 * s:16 * e].cpu().numpy()
        w = oracle.pack_round(olay, host, None, e - s, 0, P, -1, buf)
        assert s + w["next"] == int(nxt[kk]), f"round {kk}"
        assert w["send_offset"].tolist() == so[kk].tolist(), f"round {kk}"
        for p in range(P):
            o = p * stream_cap + int(base[kk][p])
            view = streams[o:o + int(so[kk][p])].cpu().numpy()
            assert np.array_equal(view, w["partitions"][p]), f"round {kk} partition {p}"
    pk.close()
    pk1.close()
