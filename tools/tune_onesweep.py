#!/usr/bin/env python3
"""Times every compiled instantiation of the one-sweep kernel, interleaved in
one process (variants x rounds), next to the plain-copy calibration of the
same tile loop.  Usage: python tools/tune_onesweep.py [records] [rounds]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from mimir_amd import capi  # noqa: E402
from mimir_amd.packer import DevicePacker, synth_u64_records  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    only = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
    P = 8
    lib = capi.load()
    names = []
    v = 0
    while lib.mgs_tuning_name(v):
        names.append(lib.mgs_tuning_name(v).decode())
        v += 1
    variants = only if only is not None else list(range(len(names)))
    lay = capi.Layout.make(capi.KEY_FIXED, 8, capi.VAL_FIXED, 8)
    buf = 16 * (n // P + n // 64 + 4096)
    rec = synth_u64_records(n, seed=42)
    pk = DevicePacker(lay, P, -1, buf, max_records=n)
    image = pk.new_send_image()
    ref = None
    times = {v: [] for v in variants}
    copies = {v: [] for v in variants}
    for rnd in range(rounds + 1):
        for v in variants:
            pk.set_tuning(v)
            r = pk.pack(rec, n, image, algo=capi.ALGO_ONESWEEP16)
            ms = pk.last_kernel_ms()
            if names[v].endswith("-stats") and rnd == rounds:
                st = pk.debug_stats()
                t = max(st[0], 1)
                tail = (f"first poll unfinished: near counts {st[5]/t:.3f} far counts {st[6]/t:.3f} "
                        f"no base {st[7]/t:.3f} per tile" if names[v][0] in "qg" else
                        f"max_depth={st[5]} max_cycles={st[6]} max_polls={st[7]}")
                print(f"[{names[v]}] tiles={st[0]} polls/tile={st[1]/t:.2f} idle/tile={st[2]/t:.2f} "
                      f"depth/tile={st[3]/t:.1f} cycles/tile={st[4]/t:.0f} {tail}")
            if "fake" not in names[v]:
                if ref is None:
                    ref = r.send_offset.copy()
                assert (r.send_offset == ref).all() and r.next == n, names[v]
            cms = pk.copy_calibration(rec, image, n)
            if rnd > 0:  # first round = warm-up
                times[v].append(ms)
                copies[v].append(cms)
    gb = n * 32 / 1e9
    print(f"n={n} records, P={P}, {rounds} timed rounds; GB/s = 32 B/record over kernel time")
    print(f"{'variant':24s} {'med ms':>8s} {'min ms':>8s} {'GB/s(med)':>10s} {'%8TB/s':>7s} | {'copy ms':>8s} {'copy GB/s':>10s}")
    for v in variants:
        med, mn = statistics.median(times[v]), min(times[v])
        cmed = statistics.median(copies[v])
        print(f"{names[v]:24s} {med:8.3f} {mn:8.3f} {gb / med * 1e3:10.0f} {gb / med * 1e3 / 80:7.1f} | "
              f"{cmed:8.3f} {gb / cmed * 1e3:10.0f}")
    pk.close()


if __name__ == "__main__":
    main()
