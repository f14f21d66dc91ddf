#!/bin/bash
# A/B on one box: the in-tree build (global instead of FLAT memory operations in the one-sweep kernels, no
# volatile LDS words, unconditional tile loads) against the build before it (mimir_amd/lib/libmgs_base.so).
# Parity of the in-tree build first; stops at the first failure.
set -u
out=gpurun_out/r2
mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_rounds.py tests/test_gpu_streams.py tests/test_zz_no_alloc.py -x -q \
    --timeout 300 --timeout-method thread > $out/memops_test.log 2>&1
rc=$?
echo rc=$rc >> $out/memops_test.log
tail -n 4 $out/memops_test.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab_bench.sh $PWD/mimir_amd/lib/libmgs_base.so 2>&1 | tee $out/ab_memops.log || exit 1
bash tools/ab_layouts.sh $PWD/mimir_amd/lib/libmgs_base.so 5e7 words,join,oct12 2>&1 | tee -a $out/ab_memops.log
