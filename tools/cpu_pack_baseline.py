#!/usr/bin/env python3
"""CPU baseline on several host cores: every process plays one Mimir rank and
runs the oracle's CollectiveShuffler::write loop (oracle/mimir_oracle.c,
mo_pack_round) over its own slice of the benchmark's <u64,u64> record stream,
packing into its own 8-partition send image.  No GPU, no torch: bench.py starts
this as a child process.  Prints one JSON line.

    python tools/cpu_pack_baseline.py --procs 16 --records 1e9
"""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _rank(i, m, P, seed, barrier, out):
    import numpy as np
    from mimir_amd import datagen
    from oracle import oracle as O
    orc = O.Oracle()
    lay = O.Layout.make(O.KEY_FIXED, 8, O.VAL_FIXED, 8)
    blob, _, _ = datagen.u64_records(m, seed=seed, first=i * m)
    buf = 16 * (m // P + m // 64 + 4096)
    img = np.zeros(P * buf, dtype=np.uint8)  # pages touched, like the reference's open()
    so = np.zeros(P, dtype=np.int64)
    lb, lr, pr = C.c_int64(), C.c_uint64(), C.c_uint64()
    barrier.wait()
    t0 = time.perf_counter()
    nxt = orc.L.mo_pack_round(C.byref(lay), blob.ctypes.data, None, 16, m, 0, P, -1, buf,
                              img.ctypes.data, so.ctypes.data, None, 0,
                              C.byref(lb), C.byref(lr), C.byref(pr))
    dt = time.perf_counter() - t0
    assert nxt == m
    out.put((i, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=16)
    ap.add_argument("--records", type=float, default=1e9, help="records over all processes")
    ap.add_argument("--seed", type=int, default=42)
    args = ap.parse_args()
    procs, P = args.procs, 8
    m = int(args.records) // procs
    ctx = mp.get_context("fork")
    barrier, out = ctx.Barrier(procs), ctx.Queue()
    ps = [ctx.Process(target=_rank, args=(i, m, P, args.seed, barrier, out)) for i in range(procs)]
    for p in ps:
        p.start()
    times = dict(out.get() for _ in ps)
    for p in ps:
        p.join()
    dt = max(times.values())
    print(json.dumps({"value": procs * m / dt, "unit": "records/s", "cores": procs, "kind": "port",
                      "GBps": procs * m * 32 / dt / 1e9, "seconds": dt,
                      "sample": f"{procs} processes x {m} consecutive <u64,u64> records of the benchmark's "
                                f"generator stream (seed {args.seed}), P=8, one round each, "
                                f"oracle/mimir_oracle.c mo_pack_round (C restatement of "
                                f"CollectiveShuffler::write), slowest process {dt:.2f} s"}), flush=True)


if __name__ == "__main__":
    main()
