#!/usr/bin/env python3
"""Device time of mgs_group_device (KV -> KMV: KMVContainer::convert) on resident
records.  Algorithmic bytes = the records read once + the values, one
group_first and one group_start entry per group written.
Usage: python tools/bench_group.py [records] [rounds] [words|u64]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mimir_amd import capi, datagen  # noqa: E402
from mimir_amd.packer import DevicePacker, group_records  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 60_000_000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    only = sys.argv[3] if len(sys.argv) > 3 else ""
    cases = []
    if only in ("", "words"):
        cases.append(("words Zipf(1.0) over 1e6, len 6+-1", datagen.word_records(n, 1_000_000, 1.0, 6, 1.0, seed=44)))
    if only in ("", "u64"):
        cases.append(("u64/u64, all keys distinct", datagen.u64_records(n, seed=42)))
    for name, (blob, offs, lay_t) in cases:
        lay = capi.Layout.make(*lay_t)
        pk = DevicePacker(lay, 2, -1, 4096, 1)
        d = torch.from_numpy(blob).cuda()
        d_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).cuda() if offs is not None else None
        stream = torch.cuda.Stream()
        ts = []
        with torch.cuda.stream(stream):
            for r in range(rounds + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                first, start, values, cuts = group_records(pk, d, n, rec_off=d_off)
                dt = time.perf_counter() - t0
                if r:
                    ts.append(dt)
                groups = int(first.numel())
                del first, start, values
        ts.sort()
        med = ts[len(ts) // 2]
        vb = lay.val_bytes
        alg = len(blob) + (4 * (n + 1) if offs is not None else 0) + n * vb + 8 * groups
        print(json.dumps({"workload": name, "records": n, "groups": groups, "bytes": len(blob),
                          "value_buffers": len(cuts) + 1, "ms": med * 1e3, "records_per_s": n / med,
                          "GBps_algorithmic": alg / med / 1e9, "frac_of_8TBps": alg / med / 8e12,
                          "note": "blocking call incl. output allocation by the wrapper"}), flush=True)
        pk.close()
        del d, d_off
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
