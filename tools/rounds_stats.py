#!/usr/bin/env python3
"""Diagnostic: where do the scan waves of the 16-byte one-sweep kernel spend their time
when a batch is cut into many rounds?  Runs the '-stats' instantiation and prints the
chained-scan counters plus the rounds counters (cycles waiting for a round's table entry,
number of such waits, cycles finding cuts).  Usage: tools/rounds_stats.py [records] [comm]"""
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
pk = DevicePacker(lay, P, -1, buf, max_records=n)
images = torch.empty(rounds * P * buf, dtype=torch.uint8, device="cuda:0")
for variant in (0, 1):
    pk.set_tuning(variant)
    for _ in range(3):
        r = pk.pack_rounds(rec, n, images, P * buf, rounds)
    st = pk.debug_stats() if variant == 1 else None
    print(json.dumps({"variant": pk.tuning_name(), "records": n, "partition_bytes": buf, "rounds": r.n_rounds,
                      "kernel_ms": pk.last_kernel_ms(),
                      "stats": None if st is None else {
                          "tiles": st[0], "polls": st[1], "idle_polls": st[2], "depth": st[3], "scan_cycles": st[4],
                          "rt_wait_cycles": st[5], "rt_waits": st[6], "cut_cycles": st[7]}}))
