"""The device-resident rows chained the way wordcount uses them
(examples/wordcount.cpp:43-58): text -> map (tokenize) -> shuffle (pack, the
exchange done by hand between P simulated ranks on one GPU) -> receive side
(unpack) -> KV->KMV (group) -> reduce (sum per key, numpy).  The word counts
must equal a plain count over the texts, every word must sit on the rank
hashlittle(word NUL) % P chose, and every step equals the oracle's."""
import collections

import numpy as np
import pytest

from mimir_amd import capi
from oracle import oracle as O

pytestmark = pytest.mark.gpu
LAY = (capi.KEY_CSTR, 0, capi.VAL_FIXED, 8)


def _text(rng, nwords, vocab):
    ranks = np.minimum(rng.zipf(1.2, size=nwords) - 1, len(vocab) - 1)
    parts = []
    for i, r in enumerate(ranks):
        parts.append(vocab[r])
        parts.append(b"\n" if i % 9 == 8 else b"  " if i % 5 == 0 else b" ")
    return b"".join(parts)


@pytest.mark.parametrize("P", [1, 4])
def test_wordcount_chain(oracle, P):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mimir_amd.packer import DevicePacker, group_records, tokenize_text, unpack_stream
    rng = np.random.default_rng(17)
    vocab = [bytes(rng.integers(97, 123, size=rng.integers(1, 10), dtype=np.uint8)) for _ in range(4000)]
    vocab = list(dict.fromkeys(vocab))
    texts = [_text(rng, 60000 + 1000 * r, vocab) for r in range(P)]
    truth = collections.Counter(w for t in texts for w in t.replace(b"\n", b" ").split(b" ") if w)
    glay, olay = capi.Layout.make(*LAY), O.Layout.make(*LAY)

    # map + pack on every simulated rank
    outgoing, local = [], []
    for r in range(P):
        text = np.frombuffer(texts[r], dtype=np.uint8)
        want_recs, want_offs = oracle.tokenize(text)
        n = len(want_offs) - 1
        buf = len(want_recs) + 16
        pk = DevicePacker(glay, P, r, buf, n)
        recs, offs, got_n, got_nb = tokenize_text(pk, torch.from_numpy(text.copy()).cuda(), len(text))
        assert (got_n, got_nb) == (n, len(want_recs))
        image = pk.new_send_image()
        loc = torch.empty(buf, dtype=torch.uint8, device="cuda")
        res = pk.pack(recs, n, image, rec_off=offs, local_out=loc)
        want = oracle.pack_round(olay, want_recs, want_offs, n, 0, P, r, buf)
        assert res.next == n and res.send_offset.tolist() == want["send_offset"].tolist()
        parts = []
        for d in range(P):
            so = int(res.send_offset[d])
            part = image[d * buf:d * buf + so]
            assert np.array_equal(part.cpu().numpy(), want["partitions"][d])
            parts.append(part)
        outgoing.append(parts)
        local.append(loc[:res.local_bytes])
        assert np.array_equal(local[-1].cpu().numpy(), want["local"])
        pk.close()

    # exchange by hand, then receive side + KV->KMV + reduce on every rank
    counted = collections.Counter()
    for d in range(P):
        stream = torch.cat([local[d]] + [outgoing[s][d] for s in range(P) if s != d])
        host = stream.cpu().numpy()
        pk = DevicePacker(glay, P, d, 4096, 1)
        rec_off, n, _ = unpack_stream(pk, stream, stream.numel())
        want_off = oracle.parse_records(olay, host)
        assert n == len(want_off) - 1
        assert np.array_equal(rec_off[:n + 1].cpu().numpy().view(np.uint32).astype(np.uint64), want_off)
        first, start, values, _ = group_records(pk, stream, n, rec_off=rec_off)
        wf, ws, wv, _ = oracle.kmv_convert(olay, host, want_off, n, hashscale=P)
        assert np.array_equal(first.cpu().numpy().view(np.uint32), wf)
        assert np.array_equal(start.cpu().numpy().view(np.uint32).astype(np.uint64), ws)
        assert np.array_equal(values.cpu().numpy(), wv)
        vals = values.cpu().numpy().view(np.uint64)
        sums = np.add.reduceat(vals, ws[:-1].astype(np.int64)) if len(wf) else []
        for g, i in enumerate(wf):
            key = bytes(host[int(want_off[i]):int(want_off[i + 1]) - 9])
            assert oracle.hashlittle(key + b"\0") % P == d
            assert key not in counted
            counted[key] = int(sums[g])
        pk.close()
    assert counted == truth
