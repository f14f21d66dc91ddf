"""Receive side on the device (mgs_unpack_device) against the CPU oracle: the
record offsets save_data's walk finds (src/collectiveshuffler.h:563-585) and
the records at which KVContainer::write opens a new page
(src/kvcontainer.h:133-160).  Integer work: exact."""
import numpy as np
import pytest

from mimir_amd import capi, datagen
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _setup(lay_t):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mimir_amd.packer import DevicePacker
    glay = capi.Layout.make(*lay_t)
    olay = O.Layout.make(*lay_t)
    pk = DevicePacker(glay, 2, -1, 4096, 1)
    return torch, pk, olay


def check(oracle, blob, lay_t, page_bytes, first_fill=0):
    from mimir_amd.packer import unpack_stream
    torch, pk, olay = _setup(lay_t)
    want_off = oracle.parse_records(olay, blob)
    want_cuts = oracle.page_cuts(want_off, page_bytes, first_fill)
    d = torch.from_numpy(np.ascontiguousarray(blob)).cuda() if len(blob) else torch.empty(1, dtype=torch.uint8, device="cuda")
    room = None  # the wrapper's own buffer has room for records of two bytes and more
    if len(want_off) + 1 > len(blob) // 2 + 2:
        room = torch.empty(len(want_off) + 1, dtype=torch.int32, device="cuda")
    rec_off, n, cuts = unpack_stream(pk, d, len(blob), page_bytes, first_fill, max_cuts=4096, rec_off=room)
    assert n == len(want_off) - 1
    got = rec_off[:n + 1].cpu().numpy().view(np.uint32).astype(np.uint64)
    assert np.array_equal(got, want_off)
    assert [c[0] for c in cuts] == want_cuts.tolist()
    assert [c[1] for c in cuts] == [int(want_off[i]) for i in want_cuts]
    pk.close()
    return n, cuts


@pytest.mark.parametrize("n", [0, 1, 2, 5, 63, 64, 65, 1000, 4097, 200000])
def test_unpack_words(oracle, n):
    blob, offs, lay = datagen.word_records(n, n_unique=700, alpha=0.7, len_mean=6, len_sd=2.5, seed=n + 1)
    check(oracle, blob, lay, 1 << 16)


def test_unpack_words_many_chunks(oracle):
    """Several chunks per stream, entry states other than "at a record start"."""
    n = 3_000_000
    blob, offs, lay = datagen.word_records(n, n_unique=5000, alpha=1.0, len_mean=5, len_sd=2, seed=5)
    got_n, cuts = check(oracle, blob, lay, 1 << 20, first_fill=12345)
    assert got_n == n and len(cuts) > 10


def test_unpack_values_full_of_zeros_and_short_keys(oracle):
    """One-character keys and all-zero values: almost every byte is a NUL."""
    rng = np.random.default_rng(2)
    n = 50000
    vocab = [bytes([c]) for c in range(97, 123)] + [b""]
    ranks = rng.integers(0, len(vocab), size=n)
    keys = [vocab[r] + b"\0" for r in ranks]
    recs = b"".join(k + b"\0" * 8 for k in keys)
    blob = np.frombuffer(recs, dtype=np.uint8)
    check(oracle, blob, (capi.KEY_CSTR, 0, capi.VAL_FIXED, 8), 4096)


@pytest.mark.parametrize("vbytes", [0, 1, 3, 16, 62])
def test_unpack_value_widths(oracle, vbytes):
    rng = np.random.default_rng(vbytes)
    n = 40000
    parts = []
    for _ in range(n):
        k = bytes(rng.integers(97, 123, size=rng.integers(0, 12), dtype=np.uint8)) + b"\0"
        v = bytes(rng.integers(0, 3, size=vbytes, dtype=np.uint8))  # many zero bytes
        parts.append(k + v)
    blob = np.frombuffer(b"".join(parts), dtype=np.uint8)
    lay = (capi.KEY_CSTR, 0, capi.VAL_FIXED, vbytes) if vbytes else (capi.KEY_CSTR, 0, capi.VAL_VOID, 0)
    check(oracle, blob, lay, 10000, first_fill=9990)


@pytest.mark.parametrize("vbytes,klen", [(0, 0), (0, 1), (1, 0), (2, 1), (6, 0)])
def test_unpack_dense_records(oracle, vbytes, klen):
    """Records of one to seven bytes: more record starts in a 64-byte block than a lane's ring of
    offsets takes between two of the wave's stores."""
    n = 300000
    rec = b"k" * klen + b"\0" + b"\0" * vbytes
    blob = np.frombuffer(rec * n, dtype=np.uint8)
    lay = (capi.KEY_CSTR, 0, capi.VAL_FIXED, vbytes) if vbytes else (capi.KEY_CSTR, 0, capi.VAL_VOID, 0)
    got_n, _ = check(oracle, blob, lay, 1 << 16, first_fill=77)
    assert got_n == n


@pytest.mark.parametrize("nbytes", [1, 2, 9, 15, 16, 17, 31, 33])
def test_unpack_streams_around_sixteen_bytes(oracle, nbytes):
    """Streams under 16 bytes take a kernel of their own."""
    blob = np.frombuffer(b"\0" * nbytes, dtype=np.uint8)  # empty keys, no value
    check(oracle, blob, (capi.KEY_CSTR, 0, capi.VAL_VOID, 0), 5)
    if nbytes % 3 == 0:
        blob = np.frombuffer(b"a\0\7" * (nbytes // 3), dtype=np.uint8)
        check(oracle, blob, (capi.KEY_CSTR, 0, capi.VAL_FIXED, 1), 7, first_fill=3)


def test_unpack_join_and_fixed(oracle):
    blob, offs, lay = datagen.join_records(100000, n_unique=2000, alpha=1.0, len_mean=6, len_sd=1, seed=45)
    check(oracle, blob, lay, 1 << 18)
    blob, _, lay = datagen.u64_records(100000, seed=3)
    check(oracle, blob, lay, 100000, first_fill=99990)
    blob, _, lay = datagen.octree_records(50000, level=11, seed=3)
    check(oracle, blob, lay, 19 * 1000)


def test_unpack_pack_round_trip(oracle):
    """Offsets found on the device feed the pack path again (second shuffle)."""
    from mimir_amd.packer import DevicePacker, unpack_stream
    import torch
    n = 60000
    blob, offs, lay = datagen.word_records(n, n_unique=900, alpha=0.9, len_mean=6, len_sd=2, seed=12)
    glay, olay = capi.Layout.make(*lay), O.Layout.make(*lay)
    pk = DevicePacker(glay, 8, 3, int(offs[-1]), n)
    d = torch.from_numpy(np.ascontiguousarray(blob)).cuda()
    rec_off, got_n, _ = unpack_stream(pk, d, len(blob))
    assert got_n == n
    image = pk.new_send_image()
    local = torch.empty(len(blob), dtype=torch.uint8, device="cuda")
    r = pk.pack(d, n, image, rec_off=rec_off, local_out=local)
    want = oracle.pack_round(olay, blob, offs, n, 0, 8, 3, int(offs[-1]))
    assert r.next == n and r.send_offset.tolist() == want["send_offset"].tolist()
    img = image.cpu().numpy()
    for p in range(8):
        so = int(want["send_offset"][p])
        assert np.array_equal(img[p * int(offs[-1]):p * int(offs[-1]) + so], want["partitions"][p])
    pk.close()


def test_unpack_errors(oracle):
    from mimir_amd.packer import unpack_stream
    torch, pk, olay = _setup((capi.KEY_CSTR, 0, capi.VAL_FIXED, 8))
    good = np.frombuffer(b"ab\0" + b"\1" * 8 + b"c\0" + b"\0" * 8, dtype=np.uint8)
    for bad in (good[:-1], good[:13], np.frombuffer(b"abc", dtype=np.uint8)):
        d = torch.from_numpy(np.ascontiguousarray(bad)).cuda()
        with pytest.raises(capi.MgsError) as ei:
            unpack_stream(pk, d, len(bad))
        assert ei.value.status == capi.ERR_MALFORMED
    d = torch.from_numpy(np.ascontiguousarray(good)).cuda()
    with pytest.raises(capi.MgsError) as ei:
        unpack_stream(pk, d, len(good), page_bytes=8)  # an 11-byte record, 8-byte pages
    assert ei.value.status == capi.ERR_RECORD_SIZE
    rec_off, n, cuts = unpack_stream(pk, d, len(good), page_bytes=11)
    assert n == 2 and cuts == [(1, 11)]
    pk.close()
    torch, pk, _ = _setup((capi.KEY_CSTR, 0, capi.VAL_CSTR, 0))
    with pytest.raises(capi.MgsError) as ei:
        unpack_stream(pk, d, len(good))
    assert ei.value.status == capi.ERR_ARG
    pk.close()


@pytest.mark.parametrize("shift_off,shift_stream", [(1, 0), (2, 3), (3, 5), (0, 9)])
def test_unpack_unaligned_buffers_and_canaries(oracle, shift_off, shift_stream):
    """Offsets buffer and stream at odd addresses; nothing is written outside
    rec_off[0..n] (the kernel stores 16 bytes at a time inside that range)."""
    from mimir_amd.packer import unpack_stream
    n = 150000
    blob, offs, lay = datagen.word_records(n, n_unique=3000, alpha=0.8, len_mean=5, len_sd=2, seed=77 + shift_off)
    torch, pk, olay = _setup(lay)
    want = oracle.parse_records(olay, blob)
    raw = torch.empty(len(blob) + 64, dtype=torch.uint8, device="cuda")
    d = raw[shift_stream:shift_stream + len(blob)]
    d.copy_(torch.from_numpy(np.ascontiguousarray(blob)))
    pad = 64
    buf = torch.full((n + 1 + 2 * pad + 4,), -7, dtype=torch.int32, device="cuda")
    view = buf[pad + shift_off:pad + shift_off + n + 1]
    rec_off, got_n, _ = unpack_stream(pk, d, len(blob), rec_off=view)
    assert got_n == n
    host = buf.cpu().numpy()
    assert np.array_equal(host[pad + shift_off:pad + shift_off + n + 1].view(np.uint32).astype(np.uint64), want)
    assert (host[:pad + shift_off] == -7).all() and (host[pad + shift_off + n + 1:] == -7).all()
    pk.close()


def test_unpack_states_that_never_meet(oracle):
    """A stream of zeros: records of an empty key and an all-zero value.  Every
    entry state stays on its own track for the whole stream."""
    n = 400000
    blob = np.zeros(9 * n, dtype=np.uint8)
    got_n, _ = check(oracle, blob, (capi.KEY_CSTR, 0, capi.VAL_FIXED, 8), 1 << 16)
    assert got_n == n
    blob = np.zeros(63 * 5000, dtype=np.uint8)
    got_n, _ = check(oracle, blob, (capi.KEY_CSTR, 0, capi.VAL_FIXED, 62), 1 << 16)
    assert got_n == 5000
