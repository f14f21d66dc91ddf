#!/usr/bin/env python3
"""Device time of mgs_combine_device (one round of the combiner shuffler: unique
keys per partition with summed values) next to mgs_pack_device on the same
records.  Usage: python tools/bench_combine.py [records] [rounds]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mimir_amd import capi, datagen  # noqa: E402
from mimir_amd.packer import DevicePacker  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 60_000_000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    P = 8
    blob, offs, lay_t = datagen.word_records(n, 1_000_000, 1.0, 6, 1.0, seed=44)
    buf = (len(blob) // 2 + 65536) // 16 * 16
    pk = DevicePacker(capi.Layout.make(*lay_t), P, -1, buf, n)
    d = torch.from_numpy(blob).cuda()
    d_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).cuda()
    image = pk.new_send_image()
    stream = torch.cuda.Stream()
    out = {"workload": "words Zipf(1.0) over 1e6, len 6+-1, 8 partitions, one round", "records": n,
           "bytes": len(blob)}
    with torch.cuda.stream(stream):
        for what in ("combine", "pack"):
            ts = []
            for r in range(rounds + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = pk.combine(d, n, image, rec_off=d_off) if what == "combine" else pk.pack(d, n, image, rec_off=d_off)
                dt = time.perf_counter() - t0
                assert res.next == n
                if r:
                    ts.append(dt)
            ts.sort()
            out[what + "_ms"] = ts[len(ts) // 2] * 1e3
            out[what + "_bytes_to_send"] = int(res.send_offset.sum())
            out[what + "_records_per_s"] = n / ts[len(ts) // 2]
            out[what + "_records_sent"] = int(res.packed_records)
    print(json.dumps(out), flush=True)
    pk.close()


if __name__ == "__main__":
    main()
