#!/bin/bash
# Kernel times of mgs_tokenize_device.  Usage (on the GPU box): bash tools/prof_unpack.sh [tag] [library]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-unp}
export MGS_LIBRARY=$2
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/$TAG
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/tools/bench_tokenize.py > $R/gpurun_out/$TAG.log 2>&1
grep workload $R/gpurun_out/$TAG.log | cut -c1-200
f=$(ls -t $(find $R/gpurun_out/$TAG -name '*kernel_stats.csv') | head -1)
cut -c1-150 $f | head -4
