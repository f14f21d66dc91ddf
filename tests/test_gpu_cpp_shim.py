"""The C++ host shim (include/mimir_gpu_shuffler.hpp) end to end: a compiled
driver runs P ranks as threads, each calling GpuShuffler<K,V>::write(key*,
val*) once per record like a Mimir map callback; what every rank's `out`
receives must equal the oracle simulation of the reference, order included."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from mimir_amd import capi, datagen
from tests import refsim
from tests.test_shuffle_gloo import make_stream

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "cpp", "shim_driver")


@pytest.mark.parametrize("device_exchange", [False, True])
@pytest.mark.parametrize("kind,P,keycount", [("u64", 2, 1), ("u64", 4, 1), ("words", 3, 1),
                                             ("oct", 4, 11), ("join", 2, 1)])
def test_cpp_shim_matches_reference(oracle, kind, P, keycount, device_exchange):
    """device_exchange: the transport moves device memory (Exchange::alltoallv_device) -- the
    partitions travel from send image to receive image on the GPU, the record boundaries of the
    receive image come from mgs_unpack_device; otherwise through host memory, as with plain MPI."""
    assert os.path.exists(DRIVER), "tests/cpp/shim_driver not built (__graft_entry__.build)"
    env = dict(os.environ, MGS_SHIM_DEVICE_EXCHANGE="1" if device_exchange else "0")
    counts = [0 if r == 1 else 5000 * (r + 1) + 11 for r in range(P)]
    comm, stage = 16384, 40000
    streams, lay = [], None
    with tempfile.TemporaryDirectory() as d:
        ind, outd = os.path.join(d, "in"), os.path.join(d, "out")
        os.makedirs(ind)
        os.makedirs(outd)
        for r in range(P):
            blob, offs, lay = make_stream(kind, r, counts[r])
            if offs is None:
                offs = datagen.fixed_offsets(counts[r], lay.fixed_len())
            streams.append((blob, offs, counts[r]))
            blob.tofile(os.path.join(ind, f"rank{r}.blob"))
            np.asarray(offs, dtype=np.uint64).tofile(os.path.join(ind, f"rank{r}.off"))
        subprocess.run([DRIVER, kind, str(P), str(comm), str(stage), ind, outd, str(keycount)],
                       check=True, timeout=240, env=env)
        buf_size = capi.geometry(comm, P)[0]
        sinks, stats = refsim.simulate_shuffle(oracle, lay, streams, P, buf_size)
        assert max(s["shuffle_times"] for s in stats) >= 3
        for r in range(P):
            data = np.fromfile(os.path.join(outd, f"rank{r}.out"), dtype=np.uint8).tobytes()
            off = np.fromfile(os.path.join(outd, f"rank{r}.outoff"), dtype=np.uint64)
            got = [data[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
            assert got == sinks[r], f"rank {r}: sink differs"
            kv, times, _ = open(os.path.join(outd, f"rank{r}.txt")).read().split()
            assert int(kv) == stats[r]["kvcount"] and int(times) == stats[r]["shuffle_times"]
            # COUNTER_SEND_BYTES / COUNTER_RECV_BYTES of the reference's profiler (src/stat.h:51-91)
            sent, recvd, dev_x = open(os.path.join(outd, f"rank{r}.counters")).read().split()[:3]
            assert int(sent) == stats[r]["send_bytes"] and int(recvd) == stats[r]["recv_bytes"]
            assert int(dev_x) == (stats[r]["shuffle_times"] if device_exchange else 0)


@pytest.mark.parametrize("kind,P,keycount", [("words", 2, 1), ("u64", 3, 1), ("join", 4, 1), ("oct", 4, 11)])
def test_cpp_nonblocking_shim_matches_reference(oracle, kind, P, keycount):
    """NBGpuShuffler<K,V> (the mirror of src/nbcollectiveshuffler.h) over exchanges
    that complete asynchronously: the same rounds, the same records from every
    source in the same order at every sink, the same counters; how the sources
    interleave is left to timing, as in the reference."""
    assert os.path.exists(DRIVER), "tests/cpp/shim_driver not built (__graft_entry__.build)"
    counts = [0 if r == 1 else 5000 * (r + 1) + 11 for r in range(P)]
    comm, stage = 16384, 40000
    streams, lay = [], None
    with tempfile.TemporaryDirectory() as d:
        ind, outd = os.path.join(d, "in"), os.path.join(d, "out")
        os.makedirs(ind)
        os.makedirs(outd)
        for r in range(P):
            blob, offs, lay = make_stream(kind, r, counts[r])
            if offs is None:
                offs = datagen.fixed_offsets(counts[r], lay.fixed_len())
            streams.append((blob, offs, counts[r]))
            blob.tofile(os.path.join(ind, f"rank{r}.blob"))
            np.asarray(offs, dtype=np.uint64).tofile(os.path.join(ind, f"rank{r}.off"))
        subprocess.run([DRIVER, kind, str(P), str(comm), str(stage), ind, outd, str(keycount), "nb"],
                       check=True, timeout=240)
        buf_size = capi.geometry(comm, P)[0]
        per, rounds, kvcount = refsim.simulate_streams(oracle, lay, streams, P, buf_size)
        assert max(len(x) for x in rounds) >= 3, "case too small to cut rounds"
        for r in range(P):
            data = np.fromfile(os.path.join(outd, f"rank{r}.out"), dtype=np.uint8).tobytes()
            off = np.fromfile(os.path.join(outd, f"rank{r}.outoff"), dtype=np.uint64)
            src = np.fromfile(os.path.join(outd, f"rank{r}.outsrc"), dtype=np.int32)
            assert len(src) == len(off) - 1
            by_src = [[] for _ in range(P)]
            for i in range(len(src)):
                by_src[int(src[i])].append(data[int(off[i]):int(off[i + 1])])
            for s in range(P):
                if by_src[s] != per[r][s]:
                    got_s, want_s = by_src[s], per[r][s]
                    first = next((i for i in range(min(len(got_s), len(want_s))) if got_s[i] != want_s[i]),
                                 min(len(got_s), len(want_s)))
                    raise AssertionError(
                        f"rank {r}: records from {s} differ: {len(got_s)} received, {len(want_s)} expected, first "
                        f"difference at record {first}: {got_s[first:first + 2]} vs {want_s[first:first + 2]}; "
                        f"same multiset: {sorted(got_s) == sorted(want_s)}")
            kv, times, buffers = open(os.path.join(outd, f"rank{r}.txt")).read().split()
            assert int(kv) == kvcount[r]
            assert 2 <= int(buffers) <= 5
            # every data round of the busiest rank is an exchange, plus at least the vote
            assert int(times) >= max(len(x) for x in rounds) + 1


@pytest.mark.parametrize("P,comm", [(2, 4096), (3, 6144), (4, 4096)])
def test_cpp_combiner_shim_matches_reference(oracle, P, comm):
    """GpuShuffler<char*, uint64_t> with combine_op = MGS_COMBINE_SUM_U64 (the mirror of
    CombineCollectiveShuffler with wordcount's combiner): what every rank's `out` receives,
    in order, and the counters equal the oracle simulation with combining rounds."""
    assert os.path.exists(DRIVER), "tests/cpp/shim_driver not built (__graft_entry__.build)"
    counts = [0 if r == 1 else 9000 * (r + 1) + 37 for r in range(P)]
    streams, lay = [], None
    with tempfile.TemporaryDirectory() as d:
        ind, outd = os.path.join(d, "in"), os.path.join(d, "out")
        os.makedirs(ind)
        os.makedirs(outd)
        for r in range(P):
            blob, offs, lay = make_stream("words", r, counts[r])
            streams.append((blob, offs, counts[r]))
            blob.tofile(os.path.join(ind, f"rank{r}.blob"))
            np.asarray(offs, dtype=np.uint64).tofile(os.path.join(ind, f"rank{r}.off"))
        subprocess.run([DRIVER, "words", str(P), str(comm), str(40000), ind, outd, "1", "sum"],
                       check=True, timeout=240)
        buf_size = capi.geometry(comm, P)[0]
        sinks, stats = refsim.simulate_shuffle(oracle, lay, streams, P, buf_size, combine_op=capi.COMBINE_SUM_U64)
        assert max(s["shuffle_times"] for s in stats) >= 3
        for r in range(P):
            data = np.fromfile(os.path.join(outd, f"rank{r}.out"), dtype=np.uint8).tobytes()
            off = np.fromfile(os.path.join(outd, f"rank{r}.outoff"), dtype=np.uint64)
            got = [data[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
            assert got == sinks[r], f"rank {r}: sink differs"
            kv, times, _ = open(os.path.join(outd, f"rank{r}.txt")).read().split()
            assert int(kv) == stats[r]["kvcount"] and int(times) == stats[r]["shuffle_times"]


@pytest.mark.parametrize("nb", [False, True])
def test_cpp_shim_send_buffer_above_one_gib(oracle, nb):
    """MIMIR_COMM_SIZE just above 1 GiB: the exchange counts in units of two bytes
    (type_log_bytes = 1, src/collectiveshuffler.h:46-53, :630-639): partitions of odd length
    travel with a byte of padding, and a source's records are found behind the padded run of
    the source before it (:582).  One round (nothing fills half a gibibyte here)."""
    assert os.path.exists(DRIVER), "tests/cpp/shim_driver not built (__graft_entry__.build)"
    P, kind = 2, "words"
    comm, stage = (1 << 30) + (1 << 21), 1 << 16
    assert capi.geometry(comm, P)[2] == 1
    counts = [4001, 2503]
    streams, lay = [], None
    with tempfile.TemporaryDirectory() as d:
        ind, outd = os.path.join(d, "in"), os.path.join(d, "out")
        os.makedirs(ind)
        os.makedirs(outd)
        for r in range(P):
            blob, offs, lay = make_stream(kind, r, counts[r])
            streams.append((blob, offs, counts[r]))
            blob.tofile(os.path.join(ind, f"rank{r}.blob"))
            np.asarray(offs, dtype=np.uint64).tofile(os.path.join(ind, f"rank{r}.off"))
        subprocess.run([DRIVER, kind, str(P), str(comm), str(stage), ind, outd, "1"] + (["nb"] if nb else []),
                       check=True, timeout=300, env=dict(os.environ, MGS_SHIM_DEVICE_EXCHANGE="0"))
        buf_size = capi.geometry(comm, P)[0]
        per, rounds, kvcount = refsim.simulate_streams(oracle, lay, streams, P, buf_size)
        odd = 0
        for r in range(P):
            data = np.fromfile(os.path.join(outd, f"rank{r}.out"), dtype=np.uint8).tobytes()
            off = np.fromfile(os.path.join(outd, f"rank{r}.outoff"), dtype=np.uint64)
            src = np.fromfile(os.path.join(outd, f"rank{r}.outsrc"), dtype=np.int32)
            by_src = [[] for _ in range(P)]
            for i in range(len(src)):
                by_src[int(src[i])].append(data[int(off[i]):int(off[i + 1])])
            for s_ in range(P):
                assert by_src[s_] == per[r][s_], f"rank {r}: records from {s_} differ"
                odd += sum(len(x) for x in per[r][s_]) & 1 if s_ != r else 0
            assert int(open(os.path.join(outd, f"rank{r}.txt")).read().split()[0]) == kvcount[r]
        assert odd > 0, "no partition of odd length: the padding was not exercised"


@pytest.mark.parametrize("device_exchange", [False, True])
@pytest.mark.parametrize("kind,P,keycount,window", [("u64", 2, 1, None), ("words", 4, 1, "40k"), ("join", 4, 1, None),
                                                    ("oct", 8, 11, None), ("u64", 8, 1, "70k"),
                                                    ("longwords", 3, 1, None)])
def test_cpp_streams_shim_matches_reference(oracle, kind, P, keycount, window, device_exchange):
    """StreamsGpuShuffler<K,V>: every rank's records go in as three device-resident batches
    (mgs_pack_streams_device chained in the same destination streams), close() exchanges the
    rounds a window at a time -- between device buffers with record boundaries from
    mgs_unpack_device, or through host mirrors -- and every sink must hold what the oracle
    simulation of the reference's shuffle gives it, in its order, with its counters.  window:
    MIMIR_GPU_WINDOW_SIZE (the default holds all rounds of these cases in one window; the small
    ones hold one or two rounds).  longwords: records the one-sweep kernels do not take."""
    assert os.path.exists(DRIVER), "tests/cpp/shim_driver not built (__graft_entry__.build)"
    env = dict(os.environ, MGS_SHIM_DEVICE_EXCHANGE="1" if device_exchange else "0")
    if window:
        env["MIMIR_GPU_WINDOW_SIZE"] = window
    counts = [0 if r == 1 else 5000 * (r % 3 + 1) + 11 * r for r in range(P)]
    comm = 16384 if P < 8 else 32768
    if kind == "longwords":
        counts, comm = [900, 0, 1400], 49152
    streams, lay = [], None
    with tempfile.TemporaryDirectory() as d:
        ind, outd = os.path.join(d, "in"), os.path.join(d, "out")
        os.makedirs(ind)
        os.makedirs(outd)
        for r in range(P):
            blob, offs, lay = make_stream(kind, r, counts[r])
            if offs is None:
                offs = datagen.fixed_offsets(counts[r], lay.fixed_len())
            streams.append((blob, offs, counts[r]))
            blob.tofile(os.path.join(ind, f"rank{r}.blob"))
            np.asarray(offs, dtype=np.uint64).tofile(os.path.join(ind, f"rank{r}.off"))
        subprocess.run([DRIVER, kind, str(P), str(comm), "0", ind, outd, str(keycount), "streams"],
                       check=True, timeout=240, env=env)
        buf_size = capi.geometry(comm, P)[0]
        sinks, stats = refsim.simulate_shuffle(oracle, lay, streams, P, buf_size)
        assert max(s["shuffle_times"] for s in stats) >= 3
        for r in range(P):
            data = np.fromfile(os.path.join(outd, f"rank{r}.out"), dtype=np.uint8).tobytes()
            off = np.fromfile(os.path.join(outd, f"rank{r}.outoff"), dtype=np.uint64)
            got = [data[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
            assert got == sinks[r], f"rank {r}: sink differs"
            kv, times, _ = open(os.path.join(outd, f"rank{r}.txt")).read().split()
            assert int(kv) == stats[r]["kvcount"] and int(times) == stats[r]["shuffle_times"]
            sent, recvd, dev_x = open(os.path.join(outd, f"rank{r}.counters")).read().split()[:3]
            assert int(sent) == stats[r]["send_bytes"] and int(recvd) == stats[r]["recv_bytes"]
            assert (int(dev_x) > 0) == device_exchange
            # the sources in the reference's order: within a round the own records, then source by source
            src = np.fromfile(os.path.join(outd, f"rank{r}.outsrc"), dtype=np.int32)
            assert len(src) == len(got)
