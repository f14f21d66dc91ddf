#!/bin/bash
# Runs on the GPU box: the default bench (with the CPU baselines), its kernel-trace stats and PMC
# traffic passes, then for the byte-granular shapes the bench lines, kernel-trace stats, SQ counter
# passes and PMC traffic.  Output under gpurun_out/r3/ ; what is judged is copied to profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python $R/bench.py > $O/final_bench.json 2> $O/final_bench.err
tail -c 1500 $O/final_bench.json
bash $R/tools/profile_bench.sh r3
bash $R/tools/bench_shapes.sh > $O/bench_shapes.log 2>&1
cat $O/bench_shapes.log
cd /tmp
for w in words join "octree --level 12"; do
  t=$(echo $w | tr -d ' -')
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$t -- \
      python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload $w > $O/bench_under_rocprof_$t.json 2> $O/stats_$t.log
  cat $O/stats_$t/*/*kernel_stats.csv | head -4
done
cd $R
for w in words join "octree --level 12"; do
  t=$(echo $w | tr -d ' -')
  echo "==== $w" >> $O/sq_counters.txt
  bash $R/tools/pmc_sq2.sh r3_$t --workload $w >> $O/sq_counters.txt 2>&1
done
cat $O/sq_counters.txt
for w in words join "octree --level 12"; do
  t=$(echo $w | tr -d ' -')
  bash $R/tools/pmc_traffic.sh r3_$t bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $w > $O/traffic_$t.txt 2>&1
  head -3 $O/traffic_$t.txt
done
# the receive-side parse: blocking call time, then its kernels
timeout -k 10 200 python3 $R/tools/bench_unpack.py 6e7 9 > $O/unpack.jsonl 2> /dev/null
cat $O/unpack.jsonl | cut -c1-220
bash $R/tools/prof_unpack.sh r3/unpack_stats > /dev/null 2>&1
cp $(ls -t $(find $O/unpack_stats -name '*kernel_stats.csv') | head -1) $O/unpack_kernel_stats.csv
head -5 $O/unpack_kernel_stats.csv
bash $R/tools/pmc_traffic.sh r3_unpack tools/bench_unpack.py 6e7 2 > $O/traffic_unpack.txt 2>&1 || true
head -6 $O/traffic_unpack.txt
bash $R/tools/pmc_sq_tool.sh r3_unpack tools/bench_unpack.py 6e7 2 > $O/unpack_sq_counters.txt 2>&1 || true
head -40 $O/unpack_sq_counters.txt
