#!/usr/bin/env python3
"""Device time of mgs_unpack_device (record offsets + page cuts of a resident
wire-form stream).  Algorithmic bytes = the stream read once + 4 bytes of
offset written per record.  Usage: python tools/bench_unpack.py [records] [rounds]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from mimir_amd import capi, datagen  # noqa: E402
from mimir_amd.packer import DevicePacker, unpack_stream  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 60_000_000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    for name, (blob, offs, lay_t) in (
            ("words V=1000 a=0.5 len=5", datagen.word_records(n, 1000, 0.5, 5, 0.0, seed=44)),
            ("u64/u64", datagen.u64_records(n, seed=42))):
        lay = capi.Layout.make(*lay_t)
        pk = DevicePacker(lay, 2, -1, 4096, 1)
        d = torch.from_numpy(blob).cuda()
        rec_off = torch.empty(n + 2, dtype=torch.int32, device="cuda")
        stream = torch.cuda.Stream()
        ts = []
        with torch.cuda.stream(stream):
            for r in range(rounds + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, got_n, cuts = unpack_stream(pk, d, len(blob), rec_off=rec_off)
                dt = time.perf_counter() - t0
                assert got_n == n
                if r:
                    ts.append(dt)
        ts.sort()
        med = ts[len(ts) // 2]
        alg = len(blob) + 4 * (n + 1)
        print(json.dumps({"workload": name, "records": n, "bytes": len(blob), "pages": len(cuts) + 1,
                          "ms": med * 1e3, "records_per_s": n / med, "GBps_algorithmic": alg / med / 1e9,
                          "frac_of_8TBps": alg / med / 8e12,
                          "note": "blocking call: kernels + one result copy + host sync"}), flush=True)
        pk.close()


if __name__ == "__main__":
    main()
