#!/usr/bin/env python3
"""One-core CPU time of the oracle's restatements of the rows either side of the
pack path (wordcount's map, the receive-side parse + page rule, KV -> KMV), on a
bounded sample of the same synthetic workloads the GPU tools use.  A reported
baseline, like bench.py's cpu_baseline.  Usage: python tools/cpu_rows_baseline.py [words]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from mimir_amd import datagen  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000
    orc = O.Oracle()
    out = []
    text, _ = datagen.word_text(n, 1_000_000, 1.0, 6, 1.0, seed=44)
    t0 = time.perf_counter()
    recs, offs = orc.tokenize(text)
    dt = time.perf_counter() - t0
    out.append({"row": "map (tokenize)", "words": n, "seconds": dt, "words_per_s": n / dt,
                "text_MBps": len(text) / dt / 1e6, "what": "oracle mo_tokenize, one core"})
    lay = O.Layout.make(O.KEY_CSTR, 0, O.VAL_FIXED, 8)
    t0 = time.perf_counter()
    po = orc.parse_records(lay, recs)
    cuts = orc.page_cuts(po, 64 << 20)
    dt = time.perf_counter() - t0
    assert len(po) - 1 == n
    out.append({"row": "receive side (parse + page cuts)", "records": n, "seconds": dt, "records_per_s": n / dt,
                "MBps": len(recs) / dt / 1e6, "what": "oracle mo_parse_records + mo_page_cuts, one core"})
    t0 = time.perf_counter()
    first, start, vals, _ = orc.kmv_convert(lay, recs, po, n, hashscale=8)
    dt = time.perf_counter() - t0
    out.append({"row": "KV -> KMV", "records": n, "groups": int(len(first)), "seconds": dt,
                "records_per_s": n / dt, "what": "oracle mo_kmv_convert (chained buckets as in HashBucket), one core"})
    for o in out:
        print(json.dumps(o), flush=True)


if __name__ == "__main__":
    main()
