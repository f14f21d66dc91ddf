"""Multi-rank host logic over gloo on CPU: staging, round cuts, the counts +
payload exchange, the -1 termination vote and the receive order, checked
against a single-process oracle simulation of the reference's shuffle.
The pack step is the oracle here (tests.refsim.OraclePacker); the GPU pack
itself is covered by test_gpu_parity.py and test_gpu_shuffle.py."""
import os
import pickle
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mimir_amd import capi, datagen
from tests import refsim


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_stream(kind, rank, n):
    if kind == "u64":
        blob, offs, lay = datagen.u64_records(n, seed=100 + rank, first=rank * 1_000_000)
    elif kind == "words":
        blob, offs, lay = datagen.word_records(n, n_unique=400, alpha=1.0, len_mean=6,
                                               len_sd=1.5, seed=7 + rank)
    elif kind == "longwords":
        # words of 100 bytes and more among short ones: a tile of them exceeds the one-sweep
        # kernels' LDS image, a record still fits a partition
        blob, offs, lay = datagen.word_records(n, n_unique=300, alpha=0.7, len_mean=110, len_sd=30, seed=17 + rank)
    elif kind == "join":
        blob, offs, lay = datagen.join_records(n, n_unique=300, alpha=0.8, len_mean=5,
                                               len_sd=1, seed=9 + rank)
    else:
        blob, offs, lay = datagen.octree_records(n, level=11, seed=3 + rank)
    return blob, offs, capi.Layout.make(*lay)


def counts_for(P):
    # ragged on purpose: one rank has nothing, one has a lot
    return [0 if r == 1 else 4000 * (r + 1) + 37 for r in range(P)]


def _worker(rank, P, port, kind, comm, outdir, use_gpu, counts, combine_op=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=P)
    try:
        from mimir_amd.shuffler import CombineGpuShuffler, GpuShuffler, ListSink, StreamsGpuShuffler
        n = counts[rank]
        blob, offs, lay = make_stream(kind, rank, n)
        sink = ListSink()
        if use_gpu == "streams":
            # device-resident batches through the streams layout, uneven on purpose
            sh = StreamsGpuShuffler(sink, lay, comm_buf_size=comm, stream_bytes=1 << 20, max_records=1 << 16)
            sh.open()
            d_blob = torch.from_numpy(np.ascontiguousarray(blob)).cuda() if n else torch.empty(16, dtype=torch.uint8,
                                                                                                device="cuda")
            d_off = None if offs is None else torch.from_numpy(offs.astype(np.uint32).view(np.int32)).cuda()
            for a, b in zip([0, n // 3], [n // 3, n]):
                if b > a:
                    fl = lay.fixed_len()
                    if fl:
                        sh.write_device_records(d_blob[a * fl:], b - a)
                    else:
                        sh.write_device_records(d_blob, b - a, d_off[a:])
            sh.close()
            with open(os.path.join(outdir, f"rank{rank}.pkl"), "wb") as f:
                pickle.dump({"records": sink.records, "kvcount": sh.get_record_count(),
                             "shuffle_times": sh.shuffle_times, "send_bytes": sh.send_bytes,
                             "recv_bytes": sh.recv_bytes}, f)
            return
        factory = None if use_gpu else refsim.OraclePacker
        if combine_op is None:
            sh = GpuShuffler(sink, lay, comm_buf_size=comm, stage_bytes=1 << 16,
                             packer_factory=factory)
        else:
            sh = CombineGpuShuffler(sink, lay, op=combine_op, comm_buf_size=comm, stage_bytes=1 << 16,
                                    packer_factory=factory)
        sh.open()
        if offs is None:
            offs = datagen.fixed_offsets(n, lay.fixed_len())
        # feed in uneven slices so that staging flushes fall inside rounds
        cuts = sorted(set([0, n // 7, n // 3, n // 2, n]))
        for a, b in zip(cuts[:-1], cuts[1:]):
            base = int(offs[a])
            sh.write_records(blob[base:int(offs[b])], offs[a:b + 1] - offs[a])
        sh.close()
        with open(os.path.join(outdir, f"rank{rank}.pkl"), "wb") as f:
            pickle.dump({"records": sink.records, "kvcount": sh.get_record_count(),
                         "shuffle_times": sh.shuffle_times, "send_bytes": sh.send_bytes,
                         "recv_bytes": sh.recv_bytes}, f)
    finally:
        dist.destroy_process_group()


def run_case(oracle, P, kind, comm, use_gpu=False, min_rounds=3, counts=None, combine_op=None):
    counts = counts or counts_for(P)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(P, _free_port(), kind, comm, d, use_gpu, counts, combine_op), nprocs=P, join=True)
        got = [pickle.load(open(os.path.join(d, f"rank{r}.pkl"), "rb")) for r in range(P)]
    streams, lay = [], None
    for r in range(P):
        blob, offs, lay = make_stream(kind, r, counts[r])
        streams.append((blob, offs, counts[r]))
    buf_size = capi.geometry(comm, P)[0]
    sinks, stats = refsim.simulate_shuffle(oracle, lay, streams, P, buf_size, combine_op=combine_op)
    assert max(s["shuffle_times"] for s in stats) >= min_rounds, "case too small to cut rounds"
    for r in range(P):
        assert got[r]["records"] == sinks[r], f"rank {r}: sink contents/order differ"
        for k in ("kvcount", "shuffle_times", "send_bytes", "recv_bytes"):
            assert got[r][k] == stats[r][k], (r, k, got[r][k], stats[r][k])
    # every record ends on the rank its key hashes to, exactly once
    total = sum(len(s) for s in sinks)
    assert total == sum(counts) or combine_op is not None


@pytest.mark.parametrize("P,kind", [(2, "u64"), (2, "words"), (3, "join"), (4, "oct"), (8, "words"), (8, "oct")])
def test_shuffle_matches_reference_simulation(oracle, P, kind):
    # (P = 8: 4 KiB partitions, the geometry step that three of the five benchmark configurations
    # take at full scale -- src/collectiveshuffler.h:33-53)
    counts = None if P < 8 else [0 if r == 1 else 1500 * (r % 3 + 1) + 37 * r for r in range(P)]
    run_case(oracle, P, kind, comm=16384 if P < 8 else 32768, counts=counts)


@pytest.mark.parametrize("P,comm", [(2, 4096), (3, 6144), (4, 4096)])
def test_combiner_shuffle_matches_reference_simulation(oracle, P, comm):
    """CombineGpuShuffler: rounds of unique keys with summed counts, cut where a new key
    does not fit, while records arrive in staging batches that end inside rounds and
    the staging buffer has to be refilled with the round's partitions."""
    counts = [0 if r == 1 else 9000 * (r + 1) + 37 for r in range(P)]
    run_case(oracle, P, "words", comm=comm, counts=counts, combine_op=capi.COMBINE_SUM_U64)


def test_exchange_layout_two_ranks(oracle):
    """exchange_round alone: receive image = src-ordered concatenation."""
    run_case(oracle, 2, "words", comm=8192)


# ---- the non-blocking shuffler (NBCollectiveShuffler mirror) --------------------------

class _TaggedSink:
    """Keeps what it is given per source rank."""

    def __init__(self, P):
        self.by_src = [[] for _ in range(P)]

    def write_records_from(self, src, blob, rec_off):
        b = blob.tobytes()
        for i in range(len(rec_off) - 1):
            self.by_src[src].append(b[int(rec_off[i]):int(rec_off[i + 1])])


def _nb_worker(rank, P, port, kind, comm, outdir, counts, use_gpu=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=P)
    try:
        from mimir_amd.shuffler import NBGpuShuffler
        n = counts[rank]
        blob, offs, lay = make_stream(kind, rank, n)
        sink = _TaggedSink(P)
        sh = NBGpuShuffler(sink, lay, comm_buf_size=comm, stage_bytes=1 << 16,
                           packer_factory=None if use_gpu else refsim.OraclePacker)
        sh.open()
        if offs is None:
            offs = datagen.fixed_offsets(n, lay.fixed_len())
        cuts = sorted(set([0, n // 7, n // 3, n // 2, n]))
        for a, b in zip(cuts[:-1], cuts[1:]):
            base = int(offs[a])
            sh.write_records(blob[base:int(offs[b])], offs[a:b + 1] - offs[a])
            sh.make_progress()
        sh.close()
        data_rounds = [r.tolist() for r in sh.round_log if int(r[0]) != -1 and int(r.sum()) > 0]
        with open(os.path.join(outdir, f"rank{rank}.pkl"), "wb") as f:
            pickle.dump({"by_src": sink.by_src, "kvcount": sh.get_record_count(),
                         "send_bytes": sh.send_bytes, "recv_bytes": sh.recv_bytes,
                         "rounds": data_rounds, "buffers": len(sh.msg)}, f)
    finally:
        dist.destroy_process_group()


def run_nb_case(oracle, P, kind, use_gpu=False):
    """Same rounds (bytes per destination, in order), same per-source record
    sequences at every sink, same counters as the blocking shuffle; only the
    interleaving of different sources is left to timing, as in the reference."""
    counts = counts_for(P)
    comm = 16384
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_nb_worker, args=(P, _free_port(), kind, comm, d, counts, use_gpu), nprocs=P, join=True)
        got = [pickle.load(open(os.path.join(d, f"rank{r}.pkl"), "rb")) for r in range(P)]
    streams, lay = [], None
    for r in range(P):
        blob, offs, lay = make_stream(kind, r, counts[r])
        streams.append((blob, offs, counts[r]))
    buf_size = capi.geometry(comm, P)[0]
    per, rounds, kvcount = refsim.simulate_streams(oracle, lay, streams, P, buf_size)
    assert max(len(x) for x in rounds) >= 3, "case too small to cut rounds"
    for r in range(P):
        for src in range(P):
            assert got[r]["by_src"][src] == per[r][src], f"rank {r}: records from {src} differ"
        assert got[r]["rounds"] == rounds[r]
        assert got[r]["kvcount"] == kvcount[r]
        assert got[r]["send_bytes"] == sum(sum(x) for x in rounds[r])
        assert got[r]["recv_bytes"] == sum(len(b) for s in range(P) if s != r for b in per[r][s])
        assert 2 <= got[r]["buffers"] <= 5


@pytest.mark.parametrize("P,kind", [(2, "words"), (3, "u64"), (4, "join"), (8, "u64")])
def test_nonblocking_shuffler_matches_reference_rounds(oracle, P, kind):
    run_nb_case(oracle, P, kind)
