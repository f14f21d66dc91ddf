"""The exchanges of a whole batch over gloo on CPU, world size 2, 3 and 8: round by round
out of destination streams (exchange_rounds with round_base) and as one all-to-all-v
(exchange_streams) or in windows while the packing goes on (StreamWindowExchange).  Every
rank's sink order, rebuilt from the received views, must equal the oracle simulation of the reference's shuffle (local records of round k, then round
k's receive image source by source, src/collectiveshuffler.h:563-585, :637-639).  The
pack step is the oracle here; the GPU packer's streams are checked against the same
oracle in test_gpu_streams.py."""
import os
import pickle
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mimir_amd import capi
from tests import refsim
from tests.test_shuffle_gloo import make_stream


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, P, port, kind, comm, counts, outdir, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=P)
    try:
        from oracle import oracle as O
        from mimir_amd.exchange import StreamWindowExchange, exchange_rounds, exchange_streams
        orc = O.Oracle()
        n = counts[rank]
        blob, offs, lay = make_stream(kind, rank, n)
        olay = refsim.olayout(lay)
        buf = capi.geometry(comm, P)[0]
        # this rank's rounds, laid out as the streams layout lays them out
        rounds, cur = [], 0
        while True:
            r = orc.pack_round(olay, blob, offs, n, cur, P, rank, buf)
            rounds.append(r)
            cur = r["next"]
            if cur >= n:
                break
        so = np.array([r["send_offset"] for r in rounds], dtype=np.int64).reshape(len(rounds), P)
        base = np.cumsum(so, axis=0) - so
        cap = int(so.sum(axis=0).max()) + 64
        streams = torch.zeros(P * cap, dtype=torch.uint8)
        for k, r in enumerate(rounds):
            for d in range(P):
                if so[k, d]:
                    o = d * cap + int(base[k, d])
                    streams[o:o + int(so[k, d])] = torch.from_numpy(np.ascontiguousarray(r["partitions"][d]))
        received = []   # per exchange: list over sources of the bytes received
        if mode == "rounds":
            stride = P * buf
            recv = torch.zeros(64 * stride, dtype=torch.uint8)
            rc, R = exchange_rounds(streams, 0, so, cap, recv, stride, round_base=base)
            for k in range(R):
                o, row = k * stride, []
                for s in range(P):
                    row.append(recv[o:o + int(rc[k, s])].numpy().tobytes())
                    o += int(rc[k, s])
                received.append(row)
        else:
            rcap = 1 << 20
            recv = torch.zeros(P * rcap, dtype=torch.uint8)
            if mode == "windows":
                # the same bytes in pieces, as a packer that is still packing releases them:
                # after "call" c the rounds cut so far are final, the open one is not
                ex = StreamWindowExchange(recv, rcap)
                sent = np.zeros(P, dtype=np.int64)
                ncalls = 3
                for c in range(ncalls):
                    upto = len(rounds) if c == ncalls - 1 else (len(rounds) * (c + 1)) // ncalls
                    closed = upto if c == ncalls - 1 else max(upto - 1, 0)   # the last round of a call is open
                    hi = base[closed - 1] + so[closed - 1] if closed > 0 else np.zeros(P, dtype=np.int64)
                    hi = np.maximum(hi, sent)
                    ex.post(streams, cap, sent, hi)
                    sent = hi.copy()
                rc, rb, R = ex.finish(so)
            else:
                rc, rb, R = exchange_streams(streams, cap, so, base, recv, rcap)
            for k in range(R):
                received.append([recv[s * rcap + int(rb[k, s]):s * rcap + int(rb[k, s]) + int(rc[k, s])].numpy().tobytes()
                                 for s in range(P)])
        # the sink order of the reference
        sink = []
        for k in range(R):
            if k < len(rounds):
                sink.extend(refsim.split_records(orc, olay, rounds[k]["local"]))
            for s in range(P):
                if s != rank and received[k][s]:
                    sink.extend(refsim.split_records(orc, olay, np.frombuffer(received[k][s], dtype=np.uint8)))
        with open(os.path.join(outdir, f"rank{rank}.pkl"), "wb") as f:
            pickle.dump({"sink": sink, "exchanges": R}, f)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["rounds", "streams", "windows"])
@pytest.mark.parametrize("P,kind", [(2, "words"), (3, "u64"), (8, "u64"), (8, "join")])
def test_batch_exchanges_give_the_reference_sink_order(oracle, P, kind, mode):
    if P == 8 and mode == "rounds" and kind == "join":
        pytest.skip("covered by the u64 case at this rank count")
    counts = [0 if r == 1 else 3000 * (r % 3 + 1) + 37 * r for r in range(P)]
    comm = 16384 if P < 8 else 32768   # (P = 8: 4 KiB partitions)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(P, _free_port(), kind, comm, counts, d, mode), nprocs=P, join=True)
        got = [pickle.load(open(os.path.join(d, f"rank{r}.pkl"), "rb")) for r in range(P)]
    streams, lay = [], None
    for r in range(P):
        blob, offs, lay = make_stream(kind, r, counts[r])
        streams.append((blob, offs, counts[r]))
    sinks, stats = refsim.simulate_shuffle(oracle, lay, streams, P, capi.geometry(comm, P)[0])
    assert max(s["shuffle_times"] for s in stats) >= 3
    for r in range(P):
        assert got[r]["exchanges"] == stats[r]["shuffle_times"]
        assert got[r]["sink"] == sinks[r], f"rank {r}: sink contents/order differ"
