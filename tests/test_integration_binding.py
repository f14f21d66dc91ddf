"""The Mimir-side binding (integration/gpushuffler.h, shown in the integration document) must be
real code: it is compiled, syntax only, against the reference's own headers
(src/baseshuffler.h:27-35, :255-265; src/interface.h:18-86), instantiated for the
record types of the examples, and the document must show the same text."""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINDING = os.path.join(ROOT, "integration", "gpushuffler.h")
REF_SRC = "/root/reference/src"
MPI_INC = "/opt/conda/include"

TU = r"""
#include "gpushuffler.h"
using namespace MIMIR_NS;
template class MIMIR_NS::GpuShuffler<char *, uint64_t>;   // wordcount
template class MIMIR_NS::GpuShuffler<uint64_t, uint64_t>;
template class MIMIR_NS::GpuShuffler<char, uint64_t>;     // octree
struct SingleVal { int tag; int64_t v; };
template class MIMIR_NS::GpuShuffler<char *, SingleVal>;  // join
template class MIMIR_NS::StreamsGpuShuffler<char *, uint64_t>;
template class MIMIR_NS::StreamsGpuShuffler<uint64_t, uint64_t>;
// a map callback that emits a device batch reaches the batch call through the Writable it is handed
void emit(Writable<uint64_t, uint64_t> *output, const void *d_records, uint64_t n)
{
    MIMIR_NS::StreamsGpuShuffler<uint64_t, uint64_t> *s =
        dynamic_cast<MIMIR_NS::StreamsGpuShuffler<uint64_t, uint64_t> *>(output);
    if (s) s->write_device_records(d_records, n);
}
void drive_streams(MPI_Comm c, Writable<uint64_t, uint64_t> *out, const void *d_records)
{
    BaseShuffler<uint64_t, uint64_t> *b =
        new MIMIR_NS::StreamsGpuShuffler<uint64_t, uint64_t>(c, out, NULL, 1, 1, false, NULL);
    b->open();
    emit(b, d_records, 1000);
    b->close();
    delete b;
}
// the class is concrete and is driven through the reference's base class
void drive(MPI_Comm c, Writable<uint64_t, uint64_t> *out)
{
    BaseShuffler<uint64_t, uint64_t> *b =
        new MIMIR_NS::GpuShuffler<uint64_t, uint64_t>(c, out, NULL, 1, 1, false, NULL);
    b->open();
    uint64_t k = 1, v = 2;
    b->write(&k, &v);
    b->make_progress();
    b->close();
    (void) b->get_record_count();
    delete b;
}
"""


def test_document_shows_the_binding_file():
    with open(BINDING) as f:
        code = f.read()
    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        assert code in f.read(), "the integration document and integration/gpushuffler.h differ"


@pytest.mark.skipif(not os.path.isdir(REF_SRC) or not os.path.exists(os.path.join(MPI_INC, "mpi.h")),
                    reason="needs the reference checkout and an mpi.h (build container only)")
def test_binding_compiles_against_reference_headers():
    gxx = shutil.which("g++")
    assert gxx
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(BINDING, os.path.join(d, "gpushuffler.h"))
        with open(os.path.join(d, "tu.cpp"), "w") as f:
            f.write(TU)
        # -fpermissive -w: the flags the reference's own sources need (oracle/Makefile)
        r = subprocess.run([gxx, "-std=c++11", "-fsyntax-only", "-fpermissive", "-w", "-I", REF_SRC,
                            "-I", MPI_INC, "-I", os.path.join(ROOT, "include"), os.path.join(d, "tu.cpp")],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-3000:]
