#!/bin/bash
# Runs on the GPU box: kernel-trace stats of the default bench, then two PMC
# passes (FETCH_SIZE and WRITE_SIZE cannot share a pass) on a short run.
# Output under gpurun_out/prof_<tag>/ ; copy what should be judged to profiles/.
set -e
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- \
    python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- \
    python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.log
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- \
    python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.log
python3 $R/tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.json
cat $OUT/stats/*/*kernel_stats.csv | head -8
cat $OUT/pmc_summary.json
