#!/usr/bin/env python3
"""Host-inclusive rate of the pack path: <u64,u64> records start in pinned host
memory and the packed partitions end in pinned host memory.

Pipeline per chunk (two slots, three streams): H2D of chunk k+1 overlaps the
pack of chunk k and the D2H of chunk k-1's partitions (only send_offset[d]
valid bytes per partition travel back, so the D2H is issued after the pack's
result is known -- as the reference needs the counts before its all-to-all-v).
Usage: python tools/bench_host_inclusive.py [records] [chunk_records]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from mimir_amd import capi  # noqa: E402
from mimir_amd.packer import DevicePacker, synth_u64_records  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 400_000_000
    c = int(float(sys.argv[2])) if len(sys.argv) > 2 else 50_000_000
    assert n % c == 0
    P = 8
    nchunks = n // c
    buf = 16 * (c // P + c // 64 + 4096)
    lay = capi.Layout.make(capi.KEY_FIXED, 8, capi.VAL_FIXED, 8)

    host_in = torch.empty(n * 16, dtype=torch.uint8, pin_memory=True)
    host_out = torch.empty(nchunks * P * buf, dtype=torch.uint8, pin_memory=True)
    tmp = torch.empty(c * 16, dtype=torch.uint8, device="cuda:0")
    for k in range(nchunks):  # seeded records, generated on the device, parked on the host
        synth_u64_records(c, seed=42, first=k * c, out=tmp)
        torch.cuda.synchronize()
        host_in[k * c * 16:(k + 1) * c * 16].copy_(tmp)
    del tmp

    d_in = [torch.empty(c * 16, dtype=torch.uint8, device="cuda:0") for _ in range(2)]
    pk = DevicePacker(lay, P, -1, buf, max_records=c)
    image = [pk.new_send_image() for _ in range(2)]
    sA, sB, sC = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    ev_h2d = [torch.cuda.Event() for _ in range(2)]
    ev_d2h = [torch.cuda.Event() for _ in range(2)]

    def h2d(k):
        slot = k % 2
        with torch.cuda.stream(sA):
            d_in[slot].copy_(host_in[k * c * 16:(k + 1) * c * 16], non_blocking=True)
            ev_h2d[slot].record(sA)

    def run():
        total_out = 0
        h2d(0)
        for k in range(nchunks):
            slot = k % 2
            sB.wait_event(ev_h2d[slot])
            sB.wait_event(ev_d2h[slot])  # image[slot] was read by the D2H of chunk k-2
            pk.pack_async(d_in[slot], c, image[slot], stream=sB.cuda_stream)
            if k + 1 < nchunks:
                h2d(k + 1)  # d_in[other slot]: its pack (chunk k-1) has been waited for below
            res = pk.result()  # host waits for the pack of chunk k
            assert res.next == c
            with torch.cuda.stream(sC):
                for b in range(P):
                    nb = int(res.send_offset[b])
                    o = (k * P + b) * buf
                    host_out[o:o + nb].copy_(image[slot][b * buf:b * buf + nb], non_blocking=True)
                    total_out += nb
                ev_d2h[slot].record(sC)
        torch.cuda.synchronize()
        return total_out

    run()  # warm-up
    t0 = time.perf_counter()
    total_out = run()
    dt = time.perf_counter() - t0
    assert total_out == n * 16
    print(json.dumps({
        "what": "host-inclusive pack: pinned host records -> H2D -> pack -> D2H of the partitions",
        "records": n, "chunk_records": c, "partitions": P, "seconds": dt,
        "records_per_s": n / dt, "h2d_GBps": n * 16 / dt / 1e9, "d2h_GBps": total_out / dt / 1e9,
        "kernel_variant": pk.tuning_name(),
    }), flush=True)
    pk.close()


if __name__ == "__main__":
    main()
