"""The rows either side of the pack path at a size where every level of their
cutting is exercised (2e7 words: thousands of chunks of 64 pieces), checked
through properties that need no oracle run of that size: two independent
kernels must agree on the record offsets (tokenizer and unpack), every record
must end with NUL + value, groups must come in first-occurrence order and add
up to the records, their number must be the number of distinct words drawn."""
import numpy as np
import pytest

from mimir_amd import capi, datagen

pytestmark = pytest.mark.gpu
LAY = (capi.KEY_CSTR, 0, capi.VAL_FIXED, 8)


def test_large_text_tokenize_unpack_group():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mimir_amd.packer import DevicePacker, group_records, tokenize_text, unpack_stream
    n = 20_000_000
    text, distinct = datagen.word_text(n, 200_000, 1.0, 6, 1.5, seed=9)
    word_bytes = int(np.count_nonzero((text != 32) & (text != 10)))
    pk = DevicePacker(capi.Layout.make(*LAY), 2, -1, 4096, 1)
    d = torch.from_numpy(text).cuda()
    recs, offs, got_n, got_nb = tokenize_text(pk, d, len(text), value=1)
    assert got_n == n and got_nb == word_bytes + 9 * n
    o = offs[:n + 1].to(torch.int64)
    assert int(o[0]) == 0 and int(o[n]) == got_nb
    lens = o[1:] - o[:-1]
    assert int(lens.min()) >= 10 and int(lens.sum()) == got_nb
    # every record ends with NUL, 1, 0 x 7
    tail = torch.tensor([0, 1, 0, 0, 0, 0, 0, 0, 0], dtype=torch.uint8, device="cuda")
    for s in range(0, n, 1 << 22):
        e = min(n, s + (1 << 22))
        at = (o[s + 1:e + 1] - 9).unsqueeze(1) + torch.arange(9, device="cuda")
        assert bool((recs[at] == tail).all().item())
    # the receive-side parse finds the same offsets in those records
    rec_off, un, cuts = unpack_stream(pk, recs, got_nb, page_bytes=64 << 20)
    assert un == n and torch.equal(rec_off[:n + 1], offs[:n + 1])
    assert len(cuts) == (got_nb - 1) // (64 << 20)  # records are far smaller than a page: every page fills up
    # KV -> KMV
    first, start, values, _ = group_records(pk, recs, n, rec_off=rec_off)
    assert first.numel() == distinct
    f, st = first.to(torch.int64), start.to(torch.int64)
    assert bool((f[1:] > f[:-1]).all().item()) and int(f[0]) == 0  # first-occurrence order
    assert int(st[0]) == 0 and int(st[-1]) == n and bool((st[1:] > st[:-1]).all().item())
    assert bool((values.view(torch.int64) == 1).all().item())
    pk.close()
