#!/usr/bin/env python3
"""Diagnostic: the 16-byte one-sweep kernel on one batch in its three forms -- one round
(partitions that hold the batch), rounds as images (cut inside the sweep), rounds as
views into destination streams (found by the trailing round finder) -- with the
finder's own clock: its wall time against the kernel's, and where its cycles go.
Usage: tools/finder_stats.py [records] [comm bytes]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mimir_amd import capi  # noqa: E402
from mimir_amd.packer import DevicePacker, synth_u64_records  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
comm = int(sys.argv[2]) if len(sys.argv) > 2 else 64 << 20
P = 8
buf = capi.geometry(comm, P)[0]
rounds = int(16 * n / P / buf * 1.15) + 8
lay = capi.Layout.make(capi.KEY_FIXED, 8, capi.VAL_FIXED, 8)
rec = synth_u64_records(n, seed=42)
cap = (16 * n // P + 16 * n // 64 + 65536) // 16 * 16
big = torch.empty(max(rounds * P * buf, P * cap), dtype=torch.uint8, device="cuda:0")

pk1 = DevicePacker(lay, P, -1, cap, max_records=n)
for _ in range(3):
    pk1.pack(rec, n, big)
print(json.dumps({"form": "one round", "records": n, "kernel_ms": pk1.last_kernel_ms()}))
pk1.close()

pk = DevicePacker(lay, P, -1, buf, max_records=n)
for _ in range(3):
    r = pk.pack_rounds(rec, n, big, P * buf, rounds)
print(json.dumps({"form": "images", "records": n, "partition_bytes": buf, "rounds": r.n_rounds,
                  "kernel_ms": pk.last_kernel_ms()}))
for _ in range(3):
    r = pk.pack_streams(rec, n, big, cap, rounds)
    ms = pk.last_kernel_ms()
    st = pk.finder_stats()
    print(json.dumps({"form": "streams", "records": n, "partition_bytes": buf, "rounds": r.n_rounds, "kernel_ms": ms,
                      "finder": {"wall_ms": st[0] / 1e5, "search_cycles_per_round": st[1] / max(r.n_rounds, 1),
                                 "hash_cycles_per_cut": st[2] / max(r.n_rounds - 1, 1),
                                 "walk_cycles_per_cut": st[3] / max(r.n_rounds - 1, 1)}}))
pk.close()
