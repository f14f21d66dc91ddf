"""The device's lookup3 against hash values produced by the reference's own
compiled hash.cpp (tests/golden/hashlittle_vectors.json): every golden key is
packed as a fixed-width record and its destination byte must equal
golden_hash % P.  The moduli are pairwise coprime and their product exceeds
2^32, so agreeing on all of them pins the full 32-bit hash (CRT), not only its
low bits.  Also the stream == NULL contract, through a plain-C driver."""
import collections
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

from mimir_amd import capi, datagen

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "hashlittle_vectors.json")


def _by_length():
    with open(GOLDEN) as f:
        vec = json.load(f)["initval0"]
    groups = collections.defaultdict(list)
    for v in vec:
        if v["len"] > 0:
            groups[v["len"]].append((bytes.fromhex(v["key_hex"]), v["hash"]))
    return groups


def _dest_on_device(lay_t, blob, n, P, algo):
    import torch
    from mimir_amd.packer import DevicePacker
    lay = capi.Layout.make(*lay_t)
    rec_len = lay.fixed_len()
    buf = (n * rec_len + 64 + 15) // 16 * 16
    pk = DevicePacker(lay, P, -1, buf, max_records=n)
    image = torch.empty(P * buf, dtype=torch.uint8, device="cuda:0")
    dest = torch.full((n,), 0xFF, dtype=torch.uint8, device="cuda:0")
    got = pk.pack(torch.from_numpy(blob).cuda(), n, image, dest_out=dest, algo=algo)
    assert got.next == n
    name = pk.last_algo()
    pk.close()
    return dest.cpu().numpy().astype(np.int64), name


# generic path: P up to 256; one-sweep kernels: P up to 64
@pytest.mark.parametrize("P,algo", [(251, capi.ALGO_GENERIC), (253, capi.ALGO_GENERIC),
                                    (255, capi.ALGO_GENERIC), (256, capi.ALGO_GENERIC),
                                    (61, capi.ALGO_AUTO), (63, capi.ALGO_AUTO), (64, capi.ALGO_AUTO),
                                    (8, capi.ALGO_AUTO), (7, capi.ALGO_AUTO)])
def test_golden_hash_vectors_on_device(P, algo):
    seen_algos = set()
    for L, items in sorted(_by_length().items()):
        reps = 40  # a few rows of records, the golden keys interleaved
        keys = [k for k, _ in items] * reps
        want = np.array([h % P for _, h in items] * reps, dtype=np.int64)
        n = len(keys)
        blob = np.frombuffer(b"".join(keys), dtype=np.uint8).copy()
        got, name = _dest_on_device((capi.KEY_FIXED, L, capi.VAL_VOID, 0), blob, n, P, algo)
        seen_algos.add(name)
        assert np.array_equal(got, want), f"key length {L}, P={P}, path {name}"
        if L < 16:
            # the same keys as the head of 16-byte records: the one-sweep kernel for 16-byte records
            rec = np.zeros((n, 16), dtype=np.uint8)
            rec[:, :L] = blob.reshape(n, L)
            rec[:, L:] = 0xA5
            got, name = _dest_on_device((capi.KEY_FIXED, L, capi.VAL_FIXED, 16 - L), rec.reshape(-1), n, P, algo)
            seen_algos.add(name)
            assert np.array_equal(got, want), f"key length {L} in 16-byte records, P={P}, path {name}"
    if algo == capi.ALGO_AUTO:
        assert {"onesweep16", "onesweepv"} <= seen_algos, seen_algos
    else:
        assert seen_algos == {"generic"}


def test_golden_hash_vectors_as_string_keys():
    """Golden keys whose only NUL is their last byte are valid char* keys (the NUL is
    hashed, src/serializer.h:464-480): packed as <char*, uint64_t> records."""
    import torch
    from mimir_amd.packer import DevicePacker
    items = [(k, h) for L, it in _by_length().items() for k, h in it
             if k.endswith(b"\0") and b"\0" not in k[:-1]]
    assert len(items) >= 8, "the golden file holds the example words with their NUL"
    reps = 50
    recs = [k + (7).to_bytes(8, "little") for k, _ in items] * reps
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint64)
    blob = np.frombuffer(b"".join(recs), dtype=np.uint8).copy()
    n = len(recs)
    for P in (61, 63, 64, 251):
        want = np.array([h % P for _, h in items] * reps, dtype=np.int64)
        lay = capi.Layout.make(capi.KEY_CSTR, 0, capi.VAL_FIXED, 8)
        buf = (len(blob) + 64) // 16 * 16
        pk = DevicePacker(lay, P, -1, buf, max_records=n)
        image = torch.empty(P * buf, dtype=torch.uint8, device="cuda:0")
        dest = torch.full((n,), 0xFF, dtype=torch.uint8, device="cuda:0")
        d_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).cuda()
        got = pk.pack(torch.from_numpy(blob).cuda(), n, image, rec_off=d_off, dest_out=dest)
        assert got.next == n
        assert np.array_equal(dest.cpu().numpy().astype(np.int64), want), f"P={P}, path {pk.last_algo()}"
        pk.close()


def _read_case(f, P):
    head = np.frombuffer(f.read(8 * (4 + P)), dtype=np.int64)
    parts = [np.frombuffer(f.read(int(head[4 + b])), dtype=np.uint8) for b in range(P)]
    local = np.frombuffer(f.read(int(head[3])), dtype=np.uint8)
    return head, parts, local


@pytest.mark.parametrize("P,rank", [(8, 3), (5, -1)])
def test_null_stream_orders_after_default_stream(oracle, P, rank):
    """A C caller copies pageable records with hipMemcpy (or writes them with a kernel on
    the default stream) and packs at once with stream == NULL: the handle's own stream
    is a blocking stream, so the pack sees the finished input."""
    from oracle import oracle as O
    driver = os.path.join(ROOT, "tests", "cpp", "null_stream_driver")
    assert os.path.exists(driver), "tests/cpp/null_stream_driver not built (__graft_entry__.build)"
    n, seed = 3_000_000, 77
    buf = 16 * (n // P + n // 50)
    blob, _, lay_t = datagen.u64_records(n, seed=5)
    blob2, _, _ = datagen.u64_records(n, seed=seed)
    olay = O.Layout.make(*lay_t)
    with tempfile.TemporaryDirectory() as d:
        src, out = os.path.join(d, "rec.bin"), os.path.join(d, "out.bin")
        blob.tofile(src)
        subprocess.run([driver, src, str(n), str(P), str(rank), str(buf), str(seed), out],
                       check=True, timeout=120)
        with open(out, "rb") as f:
            for case, data in enumerate((blob, blob2)):
                want = oracle.pack_round(olay, data, None, n, 0, P, rank, buf)
                head, parts, local = _read_case(f, P)
                assert int(head[0]) == want["next"], f"case {case}"
                assert head[4:].tolist() == want["send_offset"].tolist(), f"case {case}"
                assert int(head[2]) == want["local_records"] and int(head[3]) == want["local_bytes"]
                for b in range(P):
                    assert np.array_equal(parts[b], want["partitions"][b]), f"case {case} partition {b}"
                if rank >= 0:
                    assert np.array_equal(local, want["local"]), f"case {case} local stream"
