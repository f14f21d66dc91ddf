#!/usr/bin/env python3
"""Device time of one pack call for the record shapes other than <u64,u64>:
wordcount (NUL-terminated word + u64), join (word + 16-byte value), octree
level passes (L key chars + u64).  Inputs are resident in HBM; algorithmic
bytes = 2 * sum(record bytes) + 4 * (n + 1) offset bytes for variable width.
Usage: python tools/bench_layouts.py [records] [rounds] [words,words6,join,oct4,oct8,oct12,oct15]"""
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


def workloads(n, only):
    w = [("words", "words V=1000 a=0.5 len=5", lambda: datagen.word_records(n, 1000, 0.5, 5, 0.0, seed=44)),
         ("words6", "words V=1e6 a=1.0 len=6+-1", lambda: datagen.word_records(n, 1_000_000, 1.0, 6, 1.0, seed=44)),
         ("join", "join V=1000 (16-byte value)", lambda: datagen.join_records(n, seed=45))]
    for level in (4, 8, 12, 15):
        w.append((f"oct{level}", f"octree level {level}",
                  lambda level=level: datagen.octree_records(n, level=level, seed=43)))
    for key, name, make in w:
        if only is None or key in only:
            yield name, make()


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
    P = 8
    out = []
    for name, (blob, offs, lay_t) in workloads(n, only):
        lay = capi.Layout.make(*lay_t)
        total = int(blob.shape[0])
        buf = (total + 65536) // 16 * 16  # any skew fits: a partition can hold the whole batch
        d_blob = torch.from_numpy(blob).cuda()
        d_off = torch.from_numpy(offs.astype(np.uint32)).cuda() if offs is not None else None
        pk = DevicePacker(lay, P, -1, buf, max_records=n)
        image = pk.new_send_image()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stream = torch.cuda.Stream()  # a real stream: the library maps the null stream to its own
        ms = []
        for r in range(rounds + 1):
            torch.cuda.synchronize()
            ev0.record(stream)
            pk.pack_async(d_blob, n, image, rec_off=d_off, stream=stream.cuda_stream)
            ev1.record(stream)
            res = pk.result()
            torch.cuda.synchronize()
            assert res.next == n or os.environ.get("MGS_OSV_STATS") == "2", (name, res)
            if r:
                ms.append(ev0.elapsed_time(ev1))
        ms.sort()
        med = ms[len(ms) // 2]
        alg = 2 * total + (4 * (n + 1) if offs is not None else 0)
        kms = pk.last_kernel_ms()
        if os.environ.get("MGS_OSR_STATS") and pk.last_kernel() == "k_onesweepr":
            st = pk.debug_stats()
            tot = max(sum(st), 1)
            names = ["hash", "rank", "wait A", "store", "wait B", "reorder", "next loads + settle", "-"]
            print("# k_onesweepr worker wave 0, share of cycles: " +
                  ", ".join(f"{nm} {v / tot:.1%}" for nm, v in zip(names, st)), flush=True)
        elif os.environ.get("MGS_OSV_STATS") and pk.last_algo() == "onesweepv":
            st = pk.debug_stats()
            tot = max(sum(st), 1)
            names = ["hash", "rank", "wait A", "store", "wait B", "reorder", "wait C + park", "wait D"]
            print("# worker wave 0, share of cycles: " +
                  ", ".join(f"{nm} {v / tot:.1%}" for nm, v in zip(names, st)), flush=True)
        line = {"workload": name, "kernel_ms": kms, "records": n, "bytes": total, "algo": pk.last_algo(),
                "kernel": pk.last_kernel(), "ms": med, "records_per_s": n / med * 1e3, "GBps_algorithmic": alg / med / 1e6,
                "frac_of_8TBps": alg / med / 1e6 / 8000}
        print(json.dumps(line), flush=True)
        out.append(line)
        pk.close()
        del d_blob, d_off, image
        torch.cuda.empty_cache()


if __name__ == "__main__":
    t0 = time.time()
    main()
    print(f"# {time.time() - t0:.0f} s", flush=True)
