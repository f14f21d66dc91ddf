#!/bin/bash
# Runs on the GPU box: bench lines of the byte-granular record shapes, the layouts
# table, per-phase cycle shares and a kernel-trace of the words workload.
# Output under gpurun_out/side_<tag>/ ; copy what should be judged to profiles/.
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/side_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for w in words join octree; do
  timeout -k 10 200 python3 $R/bench.py --workload $w --records 3e7 --steps 5 --warmup 3 > $OUT/bench_$w.json 2> $OUT/bench_$w.err || exit 1
done
timeout -k 10 300 python3 $R/tools/bench_layouts.py 2e7 5 > $OUT/layouts.log 2>&1 || exit 1
grep workload $OUT/layouts.log > $OUT/layouts.jsonl
MGS_OSV_STATS=1 timeout -k 10 200 python3 $R/tools/bench_layouts.py 2e7 3 words,words6 > $OUT/phases.log 2>&1 || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- \
    python3 $R/bench.py --workload words --records 3e7 --steps 5 --warmup 2 > $OUT/bench_words_under_rocprof.json 2> $OUT/stats.log || exit 1
cat $OUT/stats/*/*kernel_stats.csv | head -6
