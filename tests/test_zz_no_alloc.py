"""mgs_pack_device_noalloc (the entry point for callers that cannot allocate)
against mgs_pack_device on the same inputs: byte-identical partitions, local
stream, destinations and round result -- on two calls in a row, the second with
a smaller input, so that what the first call left in the library's static
device array is seen not to leak into the second."""
import numpy as np
import pytest

from mimir_amd import capi, datagen

pytestmark = pytest.mark.gpu


def _both(pk, blob, offs, n, P, rank, buf, start=0):
    import torch
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(np.ascontiguousarray(blob)).to(dev)
    d_off = None if offs is None else torch.from_numpy(offs.astype(np.uint32).view(np.int32)).to(dev)
    out = []
    for fn in (pk.pack, pk.pack_noalloc):
        image = torch.full((P * buf,), 0xEE, dtype=torch.uint8, device=dev)
        local = torch.full((max(len(blob), 16),), 0xDD, dtype=torch.uint8, device=dev) if rank >= 0 else None
        dest = torch.full((max(n - start, 1),), 0xFF, dtype=torch.uint8, device=dev)
        r = fn(d_blob, n, image, start=start, rec_off=d_off, local_out=local, dest_out=dest)
        torch.cuda.synchronize()
        out.append((r, image.cpu().numpy(), None if local is None else local.cpu().numpy(), dest.cpu().numpy()))
    (ra, ia, la, da), (rb, ib, lb, db) = out
    assert (ra.next, ra.packed_records, ra.local_records, ra.local_bytes) == \
        (rb.next, rb.packed_records, rb.local_records, rb.local_bytes)
    assert ra.send_offset.tolist() == rb.send_offset.tolist()
    for p in range(P):
        so = int(ra.send_offset[p])
        assert np.array_equal(ia[p * buf:p * buf + so], ib[p * buf:p * buf + so]), f"partition {p}"
    if rank >= 0:
        assert np.array_equal(la[:ra.local_bytes], lb[:rb.local_bytes])
    assert np.array_equal(da[:ra.next - start], db[:rb.next - start])
    return ra


@pytest.mark.parametrize("kind,P,rank", [("u64", 8, -1), ("u64", 8, 3), ("words", 8, 2), ("oct12", 5, -1)])
def test_noalloc_equals_pack_device_twice(kind, P, rank):
    from mimir_amd.packer import DevicePacker
    sizes = (300_000, 70_001)  # the second call is smaller than the first

    def make(n, seed):
        if kind == "u64":
            return datagen.u64_records(n, seed=seed)
        if kind == "words":
            return datagen.word_records(n, n_unique=2000, alpha=1.0, len_mean=6, len_sd=1, seed=seed)
        return datagen.octree_records(n, level=12, seed=seed)

    from oracle import oracle as O
    orc = O.Oracle()
    _, _, lay_t = make(16, 1)
    lay = capi.Layout.make(*lay_t)
    for i, n in enumerate(sizes):
        blob, offs, _ = make(n, 10 + i)
        # first a whole batch in one round, then a partition size that cuts the round
        buf = (len(blob) // P * 2 + 4096) // 16 * 16
        pk2 = DevicePacker(lay, P, rank, buf, max_records=sizes[0])
        r = _both(pk2, blob, offs, n, P, rank, buf)
        assert r.next == n
        # ... and the no-allocation call against the oracle itself
        want = orc.pack_round(O.Layout.make(*lay_t), blob, offs, n, 0, P, rank, buf, want_bytes=False)
        assert r.next == want["next"] and r.send_offset.tolist() == want["send_offset"].tolist()
        assert (r.packed_records, r.local_records, r.local_bytes) == \
            (want["packed_records"], want["local_records"], want["local_bytes"])
        pk2.close()
        small = (len(blob) // P // 3) // 16 * 16
        pk3 = DevicePacker(lay, P, rank, small, max_records=sizes[0])
        r = _both(pk3, blob, offs, n, P, rank, small)
        assert 0 < r.next < n
        r2 = _both(pk3, blob, offs, n, P, rank, small, start=r.next)
        assert r2.next > r.next
        pk3.close()


def test_noalloc_rejects_what_it_cannot_hold():
    import torch
    from mimir_amd.packer import DevicePacker
    lay = capi.Layout.make(capi.KEY_FIXED, 8, capi.VAL_FIXED, 8)
    n = 400_000_000  # more tiles than the static scan arrays hold at 64 columns
    pk = DevicePacker(lay, 64, -1, 1 << 20, max_records=n)
    image = torch.empty(64 << 20, dtype=torch.uint8, device="cuda:0")
    rec = torch.empty(16, dtype=torch.uint8, device="cuda:0")  # never read: the call is refused first
    with pytest.raises(capi.MgsError) as e:
        pk.pack_noalloc(rec, n, image)
    assert e.value.status == capi.ERR_ARG and "static scratch" in str(e.value)
    pk.close()
    # more than 64 ranks: only the generic kernels (per-handle buffers) take that
    pk = DevicePacker(lay, 100, -1, 1 << 16, max_records=1000)
    image = torch.empty(100 << 16, dtype=torch.uint8, device="cuda:0")
    rec = torch.zeros(16000, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(capi.MgsError) as e:
        pk.pack_noalloc(rec, 1000, image)
    assert e.value.status == capi.ERR_ARG
    pk.close()


@pytest.mark.parametrize("width", [75, 76, 79])
def test_noalloc_fixed_width_limit(width):
    """Fixed widths up to 75 bytes fit the call's kernel (64 records of a wave in its LDS
    slice); wider records are refused with the limit in the message."""
    import torch
    from mimir_amd.packer import DevicePacker
    n, P, rank = 20_000, 6, 1
    rng = np.random.default_rng(width)
    blob = rng.integers(0, 256, size=n * width, dtype=np.uint8)
    lay = capi.Layout.make(capi.KEY_FIXED, 8, capi.VAL_FIXED, width - 8)
    buf = (n * width // P * 2) // 16 * 16
    pk = DevicePacker(lay, P, rank, buf, max_records=n)
    if width <= 75:
        r = _both(pk, blob, None, n, P, rank, buf)
        assert r.next == n
    else:
        image = torch.empty(P * buf, dtype=torch.uint8, device="cuda:0")
        local = torch.empty(n * width, dtype=torch.uint8, device="cuda:0")
        with pytest.raises(capi.MgsError) as e:
            pk.pack_noalloc(torch.from_numpy(blob).cuda(), n, image, local_out=local)
        assert e.value.status == capi.ERR_ARG and "75 bytes" in str(e.value)
    pk.close()
