#!/bin/bash
# A/B of two builds of the library on the receive-side parse: tools/ab_unpack.sh <other.so>; alternating, twice,
# then the kernel stats of the other build.
R=$(cd "$(dirname "$0")/.." && pwd)
for i in 1 2; do
  for lib in "" "$1"; do
    MGS_LIBRARY=$lib timeout -k 10 200 python3 $R/tools/bench_unpack.py 6e7 5 2>/dev/null | python3 -c 'import sys,json
for l in sys.stdin:
    d=json.loads(l); print("  %-14s %-28s %.4f ms  %.1f G rec/s  frac %.3f" % (sys.argv[1] or "in-tree", d["workload"], d["ms"], d["records_per_s"]/1e9, d["frac_of_8TBps"]))' "$(basename "$lib")" || exit 1
  done
done
export TMPDIR=/tmp
cd /tmp && MGS_LIBRARY=$1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2/unp_exp_stats -- python3 $R/tools/bench_unpack.py 6e7 5 > /dev/null 2>&1
cat $R/gpurun_out/r2/unp_exp_stats/*/*kernel_stats.csv | head -8
