#!/usr/bin/env python3
"""Per-record cost of the C++ shim's write(): GpuShuffler<K,V>::write -> mgs_stager_put, fed one
pair at a time like a Mimir map callback, P = 4 rank threads sharing the GPU, MIMIR_COMM_SIZE=64M,
sinks that only count.  Two staging sizes per record shape:
  'put only'  the staging buffer holds the rank's whole stream: the loop is nothing but write()
              calls (two memcpys into pinned memory and an offset), all packing happens in close();
  'amortised' a 64 MiB staging buffer: the loop also pays the uploads, pack calls and exchanges.
Beside it, on one core of the same host: the CPU port of CollectiveShuffler::write
(oracle/mimir_oracle.c mo_pack_round) over one rank's records.
Usage: tools/bench_shim_write.py [records per rank]"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from mimir_amd import capi, datagen  # noqa: E402
from oracle import oracle as O  # noqa: E402

DRIVER = os.path.join(ROOT, "tests", "cpp", "shim_driver")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
P, comm = 4, 64 << 20
orc = O.Oracle()

for kind in ("u64", "words"):
    with tempfile.TemporaryDirectory() as d:
        ind, outd = os.path.join(d, "in"), os.path.join(d, "out")
        os.makedirs(ind)
        os.makedirs(outd)
        streams = []
        for r in range(P):
            if kind == "u64":
                blob, offs, lay = datagen.u64_records(n, seed=200 + r)
                offs = datagen.fixed_offsets(n, 16)
            else:
                blob, offs, lay = datagen.word_records(n, 100_000, 1.0, 6, 1.0, seed=300 + r)
            streams.append((blob, offs))
            blob.tofile(os.path.join(ind, f"rank{r}.blob"))
            np.asarray(offs, dtype=np.uint64).tofile(os.path.join(ind, f"rank{r}.off"))
        total = max(len(b) for b, _ in streams)
        out = {"record_shape": kind, "ranks": P, "records_per_rank": n, "comm_size": comm,
               "bytes_per_record": total / n}
        for name, stage in (("put_only", total + (1 << 20)), ("amortised", 64 << 20)):
            for dev in (0, 1):
                env = dict(os.environ, MGS_SHIM_TIMING="1", MGS_SHIM_DEVICE_EXCHANGE=str(dev))
                subprocess.run([DRIVER, kind, str(P), str(comm), str(stage), ind, outd, "1"], check=True, env=env,
                               timeout=600)
                secs, sent = [], []
                for r in range(P):
                    f = open(os.path.join(outd, f"rank{r}.counters")).read().split()
                    secs.append(float(f[4]))
                    sent.append(int(f[5]))
                out[f"{name}_{'device' if dev else 'host'}_exchange"] = {
                    "staging_bytes": stage, "write_loop_s_slowest_rank": max(secs),
                    "records_per_s_per_rank": n / max(secs), "ns_per_record": max(secs) / n * 1e9,
                    "bytes_exchanged_inside_the_loop": max(sent)}
        # the CPU port of the same loop, one core, one rank's records
        blob, offs = streams[0]
        olay = O.Layout.make(*lay)
        buf = capi.geometry(comm, P)[0]
        t0 = time.perf_counter()
        cur, rounds = 0, 0
        o64 = None if kind == "u64" else np.asarray(offs, dtype=np.uint64)
        while cur < n:
            r = orc.pack_round(olay, blob, o64, n, cur, P, 0, buf)
            cur = r["next"]
            rounds += 1
        dt = time.perf_counter() - t0
        out["cpu_port_one_core"] = {"records_per_s": n / dt, "ns_per_record": dt / n * 1e9, "rounds": rounds,
                                    "what": "oracle mo_pack_round (C restatement of CollectiveShuffler::write), "
                                            "rank 0's records, incl. the Python wrapper's per-round copies"}
        print(json.dumps(out), flush=True)
