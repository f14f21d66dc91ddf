"""End-to-end shuffle with the real HIP pack path: several ranks (processes)
share cuda:0, pack on the GPU through the C ABI and exchange over gloo from
host copies of their partitions.  Every rank's sink must receive exactly what
the reference's `out` would, in the same order."""
import pytest

from tests.test_shuffle_gloo import run_case, run_nb_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,kind", [(2, "u64"), (4, "u64"), (2, "words"), (3, "join"), (4, "oct")])
def test_gpu_shuffle_matches_reference_simulation(oracle, P, kind):
    run_case(oracle, P, kind, comm=16384, use_gpu=True)


@pytest.mark.parametrize("P,kind", [(2, "u64"), (3, "words"), (4, "oct"), (2, "join")])
def test_gpu_streams_shuffler_matches_reference_simulation(oracle, P, kind):
    """StreamsGpuShuffler: device-resident batches, rounds as views into destination streams,
    one all-to-all-v for the whole map -- the sink still sees what the reference's `out` sees,
    in its order, and the counters are those of the reference's exchanges."""
    run_case(oracle, P, kind, comm=16384, use_gpu="streams")


def test_gpu_streams_shuffler_takes_long_records(oracle):
    """Words of about a hundred bytes: the one-sweep kernels refuse the streams-layout call
    (a tile's records do not fit their LDS image) and the shuffler packs the batch round by
    round with the kernels that take any record up to a partition's size, as the reference
    does -- same sinks, order and counters."""
    counts = [900, 0, 1400]
    run_case(oracle, 3, "longwords", comm=49152, use_gpu="streams", counts=counts)


def test_gpu_shuffle_default_geometry(oracle):
    """The reference's default 64 MiB buffer: buf_size 32 MiB at P = 2."""
    # about 4.5 MiB of non-local records per rank against 32 MiB partitions:
    # one data exchange, then the done vote
    run_case(oracle, 2, "u64", comm=64 << 20, use_gpu=True, min_rounds=1,
             counts=[600_000, 500_000])


@pytest.mark.parametrize("P,kind", [(2, "words"), (3, "u64")])
def test_gpu_nonblocking_shuffle(oracle, P, kind):
    """NBCollectiveShuffler mirror over the real pack path."""
    run_nb_case(oracle, P, kind, use_gpu=True)


@pytest.mark.parametrize("P,comm", [(2, 4096), (3, 6144)])
def test_gpu_combiner_shuffle(oracle, P, comm):
    """CombineGpuShuffler over the device combiner (mgs_combine_device): sinks, order and
    counters equal the oracle simulation of CombineCollectiveShuffler with a sum."""
    from mimir_amd import capi
    counts = [0 if r == 1 else 9000 * (r + 1) + 37 for r in range(P)]
    run_case(oracle, P, "words", comm=comm, use_gpu=True, counts=counts, combine_op=capi.COMBINE_SUM_U64)
