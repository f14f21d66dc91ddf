"""wordcount's map on the device (mgs_tokenize_device) against the CPU oracle:
lines as FileReader::read frames them (src/fileparser.h:21-41,
src/filereader.h:138-183), words as the map callback splits and writes them
(examples/wordcount.cpp:66-82).  Byte work: exact."""
import json
import os

import numpy as np
import pytest

from mimir_amd import capi
from oracle import oracle as O

pytestmark = pytest.mark.gpu
LAY = (capi.KEY_CSTR, 0, capi.VAL_FIXED, 8)
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _packer(P=2, buf=4096, n=1):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mimir_amd.packer import DevicePacker
    return torch, DevicePacker(capi.Layout.make(*LAY), P, -1, buf, n)


def run(text, value=1, pk=None):
    from mimir_amd.packer import tokenize_text
    torch, own = (None, None)
    if pk is None:
        torch, pk = _packer()
        own = pk
    else:
        import torch
    text = np.ascontiguousarray(text, dtype=np.uint8)
    d = torch.from_numpy(text.copy()).cuda() if len(text) else torch.empty(1, dtype=torch.uint8, device="cuda")
    recs, offs, n, nb = tokenize_text(pk, d, len(text), value)
    out = recs[:nb].cpu().numpy(), offs[:n + 1].cpu().numpy().view(np.uint32).astype(np.uint64)
    if own is not None:
        own.close()
    return out


def check(oracle, text, value=1, pk=None):
    want = oracle.tokenize(text, value)
    got = run(text, value, pk)
    assert np.array_equal(got[1], want[1])
    assert np.array_equal(got[0], want[0])
    return len(want[1]) - 1


def test_tokenize_golden(oracle):
    with open(os.path.join(GOLD, "tokenize_vectors.json")) as f:
        gold = json.load(f)["vectors"]
    torch, pk = _packer()
    for v in gold:
        text = np.frombuffer(bytes.fromhex(v["text_hex"]), dtype=np.uint8)
        recs, offs = run(text, 1, pk)
        assert bytes(recs).hex() == v["records_hex"]
        assert [int(x) for x in offs] == v["rec_off"]
    pk.close()


def _random_text(rng, n, alphabet):
    a = np.frombuffer(alphabet, dtype=np.uint8)
    return a[rng.integers(0, len(a), size=n)]


@pytest.mark.parametrize("n", [0, 1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097,
                               8191, 8192, 8193, 16383, 16384, 16385, 100000])
def test_tokenize_sizes(oracle, n):
    rng = np.random.default_rng(n)
    check(oracle, _random_text(rng, n, b"abcdefghijklmnopqrstuvwxyz     \n"))
    check(oracle, _random_text(rng, n, b"ab \n\0"), value=0x0102030405060708)


def test_tokenize_corpus_many_chunks(oracle):
    """A Zipf word corpus with lines, several hundred chunks."""
    rng = np.random.default_rng(3)
    vocab = [bytes(rng.integers(97, 123, size=rng.integers(1, 12), dtype=np.uint8)) for _ in range(5000)]
    ranks = np.minimum(rng.zipf(1.3, size=600000) - 1, len(vocab) - 1)
    parts = []
    for i, r in enumerate(ranks):
        parts.append(vocab[r])
        parts.append(b"\n" if i % 11 == 10 else b" ")
    text = np.frombuffer(b"".join(parts), dtype=np.uint8)
    n = check(oracle, text)
    assert n == len(ranks)
    check(oracle, text[:-1])  # the last word ends with the text


@pytest.mark.parametrize("L", [63, 64, 65, 127, 128, 129, 1000, 1022, 1023, 1024, 1025, 2047, 2048, 5000, 70000])
def test_tokenize_long_words(oracle, L):
    """Words that cover whole 64-byte blocks, around the 1024-byte rule, at
    every kind of position relative to blocks, pieces and chunks."""
    rng = np.random.default_rng(L)
    parts = []
    for lead in (0, 1, 5, 63, 64, 200, 255, 256, 257, 4095, 16380):
        parts.append(_random_text(rng, lead, b"abc \n").tobytes())
        parts.append(b" " + b"x" * L + (b"\n" if lead % 2 else b" "))
        parts.append(b"y" * (L - 1) + b" ")
    text = np.frombuffer(b"".join(parts) + b"z" * L, dtype=np.uint8)  # the last one ends with the text
    check(oracle, text)
    check(oracle, np.frombuffer(b"q" * L, dtype=np.uint8))


@pytest.mark.parametrize("start", [0, 1, 15, 16, 17, 1000, 1023, 1024, 1025, 2047, 3000, 4095, 4096, 4097, 5119, 8191])
def test_tokenize_word_length_rule_at_step_and_chunk_edges(oracle, start):
    """Words of 1022..1025 and 2048..2049 bytes that begin at every kind of offset inside a
    1 KiB step of a wave and around the 4 KiB chunk edges: the length of a word is known from
    this step's and the next step's masks, or from the text where a chunk begins inside it."""
    torch, pk = _packer()
    for L in (1, 1022, 1023, 1024, 1025, 2047, 2048, 2049, 3071, 3072, 4096, 9000):
        head = (b"ab " * 4000)[:start]
        if head and head[-1:] != b" ":
            head = head[:-1] + b" "
        for tail in (b"", b" ", b" cd ef\n", b"\0g"):
            check(oracle, np.frombuffer(head + b"w" * L + tail, dtype=np.uint8), pk=pk)
    pk.close()


def test_tokenize_only_delimiters_and_only_words(oracle):
    check(oracle, np.full(50000, ord(" "), dtype=np.uint8))
    check(oracle, np.full(50000, ord("\n"), dtype=np.uint8))
    check(oracle, np.zeros(50000, dtype=np.uint8))
    check(oracle, np.frombuffer(b"a " * 40000, dtype=np.uint8))
    check(oracle, np.frombuffer(b"a\n" * 40000 + b"b", dtype=np.uint8))


@pytest.mark.parametrize("shift_text,shift_out,shift_off", [(1, 3, 1), (7, 0, 2), (0, 9, 3), (13, 15, 0)])
def test_tokenize_unaligned_buffers_and_canaries(oracle, shift_text, shift_out, shift_off):
    from mimir_amd.packer import tokenize_text
    torch, pk = _packer()
    rng = np.random.default_rng(shift_text)
    text = _random_text(rng, 300000, b"abcdefghij   \n")
    want_recs, want_offs = oracle.tokenize(text, 5)
    n, nb = len(want_offs) - 1, len(want_recs)
    raw = torch.empty(len(text) + 64, dtype=torch.uint8, device="cuda")
    d = raw[shift_text:shift_text + len(text)]
    d.copy_(torch.from_numpy(text.copy()))
    pad = 256
    obuf = torch.full((nb + 2 * pad + 32,), 0xEE, dtype=torch.uint8, device="cuda")
    oview = obuf[pad + shift_out:pad + shift_out + nb]  # exactly the bytes needed
    fbuf = torch.full((n + 1 + 2 * pad + 8,), -7, dtype=torch.int32, device="cuda")
    fview = fbuf[pad + shift_off:pad + shift_off + n + 1]
    _, _, got_n, got_nb = tokenize_text(pk, d, len(text), 5, records=oview, rec_off=fview)
    assert (got_n, got_nb) == (n, nb)
    oh, fh = obuf.cpu().numpy(), fbuf.cpu().numpy()
    assert np.array_equal(oh[pad + shift_out:pad + shift_out + nb], want_recs)
    assert (oh[:pad + shift_out] == 0xEE).all() and (oh[pad + shift_out + nb:] == 0xEE).all()
    assert np.array_equal(fh[pad + shift_off:pad + shift_off + n + 1].view(np.uint32).astype(np.uint64), want_offs)
    assert (fh[:pad + shift_off] == -7).all() and (fh[pad + shift_off + n + 1:] == -7).all()
    # one byte or one entry short: refused, nothing written past the end
    with pytest.raises(capi.MgsError) as ei:
        tokenize_text(pk, d, len(text), 5, records=obuf[pad:pad + nb - 1], rec_off=fview)
    assert ei.value.status == capi.ERR_ARG
    with pytest.raises(capi.MgsError) as ei:
        tokenize_text(pk, d, len(text), 5, records=oview, rec_off=fbuf[pad:pad + n])
    assert ei.value.status == capi.ERR_ARG
    pk.close()


def test_tokenize_then_pack(oracle):
    """The records feed the pack path as they are: text -> partitions."""
    from mimir_amd.packer import DevicePacker, tokenize_text
    import torch
    rng = np.random.default_rng(11)
    text = _random_text(rng, 400000, b"abcdefgh      \n")
    want_recs, want_offs = oracle.tokenize(text)
    n = len(want_offs) - 1
    P, buf = 8, len(want_recs)
    pk = DevicePacker(capi.Layout.make(*LAY), P, 3, buf, n)
    d = torch.from_numpy(text.copy()).cuda()
    recs, offs, got_n, got_nb = tokenize_text(pk, d, len(text))
    assert got_n == n
    image = pk.new_send_image()
    local = torch.empty(buf, dtype=torch.uint8, device="cuda")
    r = pk.pack(recs, n, image, rec_off=offs, local_out=local)
    want = oracle.pack_round(O.Layout.make(*LAY), want_recs, want_offs, n, 0, P, 3, buf)
    assert r.next == n and r.send_offset.tolist() == want["send_offset"].tolist()
    img = image.cpu().numpy()
    for p in range(P):
        so = int(want["send_offset"][p])
        assert np.array_equal(img[p * buf:p * buf + so], want["partitions"][p])
    assert np.array_equal(local[:r.local_bytes].cpu().numpy(), want["local"])
    pk.close()


def test_tokenize_wrong_layout():
    from mimir_amd.packer import DevicePacker, tokenize_text
    import torch
    pk = DevicePacker(capi.Layout.make(capi.KEY_FIXED, 8, capi.VAL_FIXED, 8), 2, -1, 4096, 1)
    d = torch.zeros(64, dtype=torch.uint8, device="cuda")
    with pytest.raises(capi.MgsError) as ei:
        tokenize_text(pk, d, 64)
    assert ei.value.status == capi.ERR_ARG
    pk.close()
