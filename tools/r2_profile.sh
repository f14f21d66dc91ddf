#!/bin/bash
# Runs on the GPU box: parity of the streams layout, the finder's clock, then the profiles of the default
# bench (kernel-trace stats, two PMC passes for the HBM traffic) and of the words workload.
set -e
mkdir -p gpurun_out/r2
timeout -k 10 300 python -u -m pytest tests/test_gpu_streams.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r2/streams_test.log 2>&1
timeout -k 10 200 python tools/finder_stats.py > gpurun_out/r2/finder_stats.jsonl 2> gpurun_out/r2/finder_stats.err
cat gpurun_out/r2/finder_stats.jsonl
timeout -k 10 200 python tools/finder_stats.py 1e9 16777216 > gpurun_out/r2/finder_stats_16m.jsonl 2>> gpurun_out/r2/finder_stats.err
tail -n 1 gpurun_out/r2/finder_stats_16m.jsonl
bash tools/profile_bench.sh r2
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2_words/stats -- \
    python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload words > $R/gpurun_out/prof_r2_words/bench_under_rocprof.json 2> $R/gpurun_out/prof_r2_words/stats.log
cat $R/gpurun_out/prof_r2_words/stats/*/*kernel_stats.csv | head -6
