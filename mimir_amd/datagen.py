"""Seeded synthetic record streams in wire form (numpy, host side).

The reference's generators seed from std::random_device
(generator/gen_words.cpp:85-86, generator/common.cpp:55-56) and are therefore
not reproducible; these mirror their distributions with explicit seeds.
Every function returns (blob uint8[...], rec_off uint64[n+1] or None, layout
tuple) where layout = (key_kind, key_bytes, val_kind, val_bytes).
"""
import numpy as np

KEY_FIXED, KEY_CSTR = 0, 1
VAL_VOID, VAL_FIXED, VAL_CSTR = 0, 1, 2

_ALPHABET = np.frombuffer(
    b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789",
    dtype=np.uint8)  # generator/common.cpp:34-39


def splitmix64(seed, counters):
    """splitmix64 output for state = seed + (counter+1)*golden (vectorised)."""
    with np.errstate(over="ignore"):
        c = np.asarray(counters, dtype=np.uint64)
        z = np.uint64(seed) + (c + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def u64_records(n, seed=42, first=0):
    """<uint64_t,uint64_t> records: key = splitmix64(seed, i), val = i.
    Same stream as the device generator mgs_synth_u64_records."""
    idx = np.arange(first, first + n, dtype=np.uint64)
    rec = np.empty((n, 2), dtype=np.uint64)
    rec[:, 0] = splitmix64(seed, idx)
    rec[:, 1] = idx
    return rec.view(np.uint8).reshape(-1), None, (KEY_FIXED, 8, VAL_FIXED, 8)


def unique_words(n_unique, len_mean=5, len_sd=0.0, seed=44):
    """Vocabulary of distinct words over the 62-character alphabet, length
    uniform-int in [max(0, mean-3sd), mean+3sd] (generator/common.cpp:53-95).
    Zero-length draws are skipped (strtok never yields an empty word)."""
    rng = np.random.default_rng(seed)
    lo = int(max(0.0, len_mean - 3 * len_sd))
    hi = int(len_mean + 3 * len_sd)
    lo = max(lo, 1)
    words, seen = [], set()
    while len(words) < n_unique:
        L = int(rng.integers(lo, hi + 1))
        w = _ALPHABET[rng.integers(0, 62, size=L)].tobytes()
        if w not in seen:
            seen.add(w)
            words.append(w)
    return words


def zipf_ranks(n, n_unique, alpha, rng):
    """n draws from Zipf(alpha) over ranks 0..n_unique-1 by inverting the CDF
    sum 1/i^alpha (generator/common.cpp:154-183)."""
    w = 1.0 / np.power(np.arange(1, n_unique + 1, dtype=np.float64), alpha)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    return np.minimum(np.searchsorted(cdf, rng.random(n), side="left"),
                      n_unique - 1)


def _assemble(keys, vals_bytes):
    """keys: list[bytes] (with NUL), vals_bytes: uint8[n, vb] -> blob, offs."""
    n = len(keys)
    vb = vals_bytes.shape[1] if vals_bytes is not None else 0
    lens = np.fromiter((len(k) for k in keys), dtype=np.int64, count=n) + vb
    offs = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lens, out=offs[1:].view(np.int64))
    blob = np.empty(int(offs[-1]), dtype=np.uint8)
    pos = 0
    for i, k in enumerate(keys):
        kl = len(k)
        blob[pos:pos + kl] = np.frombuffer(k, dtype=np.uint8)
        if vb:
            blob[pos + kl:pos + kl + vb] = vals_bytes[i]
        pos += kl + vb
    return blob, offs


def word_records(n, n_unique=1000, alpha=0.5, len_mean=5, len_sd=0.0, seed=44):
    """wordcount map output <char*,uint64_t>: word, NUL, u64(1)
    (examples/wordcount.cpp:48-49,72-79; test setting check-wc.sh:22-23)."""
    rng = np.random.default_rng(seed)
    vocab = unique_words(n_unique, len_mean, len_sd, seed)
    ranks = zipf_ranks(n, n_unique, alpha, rng)
    return word_records_from(vocab, ranks)


def word_records_from(vocab, ranks):
    """Vectorised assembly of word||NUL||u64(1) records for vocab[ranks]."""
    vlen = np.fromiter((len(w) for w in vocab), dtype=np.int64,
                       count=len(vocab))
    vmax = int(vlen.max())
    table = np.zeros((len(vocab), vmax + 9), dtype=np.uint8)
    for i, w in enumerate(vocab):
        table[i, :len(w)] = np.frombuffer(w, dtype=np.uint8)
        table[i, len(w) + 1] = 1  # NUL at len(w), then LE u64 1
    rl = vlen[ranks] + 9
    offs = np.zeros(len(ranks) + 1, dtype=np.uint64)
    np.cumsum(rl, out=offs[1:].view(np.int64))
    rows = table[ranks]
    mask = np.arange(vmax + 9)[None, :] < rl[:, None]
    blob = rows[mask]
    return blob, offs, (KEY_CSTR, 0, VAL_FIXED, 8)


def word_text(n, n_unique=1000, alpha=0.5, len_mean=5, len_sd=0.0, per_line=11, seed=44):
    """The text wordcount reads (generator/gen_words-like): the words of
    word_records(...) with a space after each, every `per_line`-th followed by
    a newline instead.  Returns (text uint8, number of distinct words used)."""
    rng = np.random.default_rng(seed)
    vocab = unique_words(n_unique, len_mean, len_sd, seed)
    ranks = zipf_ranks(n, n_unique, alpha, rng)
    blob, offs, _ = word_records_from(vocab, ranks)
    ends = offs[1:].astype(np.int64)
    keep = np.ones(len(blob), dtype=bool)
    for k in range(1, 9):  # drop the 8 value bytes, the key's NUL becomes the separator
        keep[ends - k] = False
    text = blob.copy()
    nul = ends - 9
    text[nul] = ord(" ")
    text[nul[per_line - 1::per_line]] = ord("\n")
    return text[keep], int(len(np.unique(ranks)))


def join_records(n, n_unique=1000, alpha=0.5, len_mean=5, len_sd=0.0,
                 val_len=4, tag=0xAA, seed=45):
    """join map output <char*,SingleVal>: word, NUL, {int32 tag; 4 pad bytes
    (zero here); int64 value} (examples/join.cpp:21-43,110-117); value
    uniform in [10^(L-1), 10^L-1] (generator/gen_words.cpp:432-447)."""
    rng = np.random.default_rng(seed)
    vocab = unique_words(n_unique, len_mean, len_sd, seed)
    ranks = zipf_ranks(n, n_unique, alpha, rng)
    vals = np.zeros((n, 16), dtype=np.uint8)
    vals[:, 0:4] = np.full(n, tag, dtype="<i4").view(np.uint8).reshape(n, 4)
    v = rng.integers(10 ** (val_len - 1), 10 ** val_len, size=n, dtype=np.int64)
    vals[:, 8:16] = v.astype("<i8").view(np.uint8).reshape(n, 8)
    keys = [vocab[r] + b"\0" for r in ranks]
    blob, offs = _assemble(keys, vals)
    return blob, offs, (KEY_CSTR, 0, VAL_FIXED, 16)


def octkeys(points, digits=15, lo=-4.0, hi=4.0):
    """Octant key of every 3-D point: `digits` characters '0'..'7', bit 0 = x,
    bit 1 = y, bit 2 = z above the running midpoint
    (examples/octree_clustering.cpp:266-304)."""
    n = points.shape[0]
    mn = np.full((n, 3), lo, dtype=np.float64)
    mx = np.full((n, 3), hi, dtype=np.float64)
    out = np.empty((n, digits), dtype=np.uint8)
    for c in range(digits):
        mid = mn + (mx - mn) / 2
        above = points > mid
        mn = np.where(above, mid, mn)
        mx = np.where(above, mx, mid)
        out[:, c] = (ord("0") + above[:, 0] + 2 * above[:, 1]
                     + 4 * above[:, 2]).astype(np.uint8)
    return out


def octree_records(n, level=8, seed=43, sigma=0.5):
    """octree level pass <char,uint64_t>, keycount = level: first `level`
    characters of the 15-digit key, then u64(1)
    (examples/octree_clustering.cpp:98-106,226-235); points ~ N(0, sigma)
    clipped into the (-4,4] box (generator/gen_3d_points.cpp:91-131)."""
    rng = np.random.default_rng(seed)
    pts = np.clip(rng.normal(0.0, sigma, size=(n, 3)), -3.999999, 4.0)
    keys = octkeys(pts)
    rec = np.zeros((n, level + 8), dtype=np.uint8)
    rec[:, :level] = keys[:, :level]
    rec[:, level] = 1
    return rec.reshape(-1), None, (KEY_FIXED, level, VAL_FIXED, 8)


def fixed_offsets(n, rec_len):
    return np.arange(n + 1, dtype=np.uint64) * np.uint64(rec_len)
