#!/usr/bin/env python3
"""The device-resident stretch of a wordcount job on one GPU, stage by stage:
text -> map (tokenize) -> shuffle pack into P partitions (plain, and with the
sum combiner) -> receive side (offsets + page cuts of every partition, as if it
had arrived) -> KV->KMV of every partition.  Wall time per stage (blocking
calls, medians) and words/s of the whole stretch.
Usage: python tools/bench_chain.py [words] [rounds]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from mimir_amd import capi, datagen  # noqa: E402
from mimir_amd.packer import DevicePacker, group_records, tokenize_text, unpack_stream  # noqa: E402


def timed(fn, rounds):
    ts, out = [], None
    for r in range(rounds + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        if r:
            ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3, out


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 60_000_000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    P = 8
    text, distinct = datagen.word_text(n, 1_000_000, 1.0, 6, 1.0, seed=44)
    d_text = torch.from_numpy(text).cuda()
    nbytes_rec = len(text) + 9 * n
    buf = (nbytes_rec // 2 + 65536) // 16 * 16  # Zipf keys: room for a hot partition
    pk = DevicePacker(capi.Layout.make(capi.KEY_CSTR, 0, capi.VAL_FIXED, 8), P, -1, buf, n)
    records = torch.empty(nbytes_rec + 64, dtype=torch.uint8, device="cuda")
    rec_off = torch.empty(n + 2, dtype=torch.int32, device="cuda")
    image = pk.new_send_image()
    stream = torch.cuda.Stream()
    res = {"workload": "text of Zipf(1.0) words over 1e6, len 6+-1, 11 per line; 8 partitions", "words": n,
           "text_bytes": len(text), "distinct": distinct}
    with torch.cuda.stream(stream):
        res["tokenize_ms"], tok = timed(lambda: tokenize_text(pk, d_text, len(text), 1, records=records, rec_off=rec_off), rounds)
        assert tok[2] == n
        res["combine_ms"], cr = timed(lambda: pk.combine(records, n, image, rec_off=rec_off), rounds)
        assert cr.next == n
        res["combined_bytes"] = int(cr.send_offset.sum())
        res["pack_ms"], pr = timed(lambda: pk.pack(records, n, image, rec_off=rec_off), rounds)
        assert pr.next == n
        parts = [image[p * buf:p * buf + int(pr.send_offset[p])] for p in range(P)]
        offs = [torch.empty(n // 2 + 2, dtype=torch.int32, device="cuda") for _ in range(P)]

        def unpack_all():
            return [unpack_stream(pk, parts[p], parts[p].numel(), rec_off=offs[p]) for p in range(P)]
        res["unpack_ms"], un = timed(unpack_all, rounds)
        assert sum(u[1] for u in un) == n

        def group_all():
            return [group_records(pk, parts[p], un[p][1], rec_off=offs[p])[0].numel() for p in range(P)]
        res["group_ms"], groups = timed(group_all, rounds)
        assert sum(groups) == distinct
    total = res["tokenize_ms"] + res["pack_ms"] + res["unpack_ms"] + res["group_ms"]
    res["stretch_ms"] = total
    res["words_per_s"] = n / total * 1e3
    res["note"] = "blocking calls one after the other on one GPU; the exchange between pack and unpack is not included"
    print(json.dumps(res), flush=True)
    pk.close()


if __name__ == "__main__":
    main()
