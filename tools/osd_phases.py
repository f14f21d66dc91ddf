#!/usr/bin/env python3
"""Diagnostic: per-phase cycle counts of the byte-granular one-sweep kernel's worker
waves (MGS_OSD_STATS=1 build), for the words / join / octree workloads.
Usage: tools/osd_phases.py [records]"""
import json
import os
import sys

MODE = os.environ.setdefault("MGS_OSD_STATS", "1")  # 2: the scan wave's phases
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mimir_amd import capi, datagen  # noqa: E402
from mimir_amd.packer import DevicePacker  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 30_000_000
names = ["hash", "rank+publish", "barrier", "counts+loads", "store", "rotate", "park+records+offsets"]
for wl in ("words", "join", "octree"):
    if wl == "words":
        blob, offs, lt = datagen.word_records(n, 1_000_000, 1.0, 6, 1.0, seed=44)
    elif wl == "join":
        blob, offs, lt = datagen.join_records(n, seed=45)
    else:
        blob, offs, lt = datagen.octree_records(n, level=12, seed=43)
    P = 8
    buf = (len(blob) // 2 + 65536) // 16 * 16
    pk = DevicePacker(capi.Layout.make(*lt), P, -1, buf, max_records=n)
    image = torch.empty(P * buf, dtype=torch.uint8, device="cuda:0")
    d_blob = torch.from_numpy(blob).cuda()
    d_off = None if offs is None else torch.from_numpy(offs.astype(np.uint32).view(np.int32)).cuda()
    for _ in range(3):
        r = pk.pack(d_blob, n, image, rec_off=d_off)
    st = pk.debug_stats()
    if MODE == "2":
        print(json.dumps({"workload": wl, "records": n, "kernel_ms": pk.last_kernel_ms(), "scan_wave": {
            "tiles": st[3], "cycles_per_tile": {"barrier": round(st[0] / max(st[3], 1)), "chained_scan": round(st[1] / max(st[3], 1)),
                                               "rest": round(st[2] / max(st[3], 1))},
            "polls_per_tile": st[4] / max(st[3], 1), "idle_polls_per_tile": st[5] / max(st[3], 1),
            "depth_per_tile": st[6] / max(st[3], 1)}}))
        pk.close()
        continue
    tot = sum(st[:7])
    print(json.dumps({"workload": wl, "records": n, "bytes_per_record": len(blob) / n, "kernel_ms": pk.last_kernel_ms(),
                      "trips_of_wave0_all_workgroups": st[7],
                      "cycles_per_trip": {nm: round(st[i] / max(st[7], 1)) for i, nm in enumerate(names)},
                      "share": {nm: round(st[i] / tot, 3) for i, nm in enumerate(names)}}))
    pk.close()
