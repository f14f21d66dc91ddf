#!/usr/bin/env python3
"""Device time of mgs_tokenize_device (wordcount's map: text -> <word,1> records
+ offsets) on a resident synthetic corpus: Zipf words, 11 per line.
Algorithmic bytes = the text read once + the records and 4 bytes of offset per
record written.  Usage: python tools/bench_tokenize.py [words] [rounds]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mimir_amd import capi, datagen  # noqa: E402
from mimir_amd.packer import DevicePacker, tokenize_text  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 60_000_000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    text, _ = datagen.word_text(n, 1_000_000, 1.0, 6, 1.0, seed=44)
    pk = DevicePacker(capi.Layout.make(capi.KEY_CSTR, 0, capi.VAL_FIXED, 8), 2, -1, 4096, 1)
    d = torch.from_numpy(text).cuda()
    records = torch.empty(len(text) + 9 * n + 64, dtype=torch.uint8, device="cuda")
    rec_off = torch.empty(n + 2, dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream()
    ts = []
    with torch.cuda.stream(stream):
        for r in range(rounds + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, _, got_n, got_nb = tokenize_text(pk, d, len(text), 1, records=records, rec_off=rec_off)
            dt = time.perf_counter() - t0
            assert got_n == n, (got_n, n)
            if r:
                ts.append(dt)
    ts.sort()
    med = ts[len(ts) // 2]
    alg = len(text) + got_nb + 4 * (n + 1)
    print(json.dumps({"workload": "text of Zipf(1.0) words over 1e6, len 6+-1, 11 per line", "words": n,
                      "text_bytes": len(text), "record_bytes": got_nb, "ms": med * 1e3,
                      "words_per_s": n / med, "text_GBps": len(text) / med / 1e9,
                      "GBps_algorithmic": alg / med / 1e9, "frac_of_8TBps": alg / med / 8e12,
                      "note": "blocking call: kernels + one result copy + host sync"}), flush=True)
    pk.close()


if __name__ == "__main__":
    main()
