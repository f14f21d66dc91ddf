#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_layouts.sh <other.so> [records] [workloads]
# Runs tools/bench_layouts.py with the in-tree build and with the other one, alternating, twice.
R=$(cd "$(dirname "$0")/.." && pwd)
OTHER=$1; N=${2:-2e7}; W=${3:-words,words6,join,oct12}
for i in 1 2; do
  echo "# in-tree build, pass $i"
  timeout -k 10 200 python3 $R/tools/bench_layouts.py $N 7 $W | grep workload | python3 -c 'import sys,json
for l in sys.stdin:
    d=json.loads(l); print("  %-32s %.4f ms  %.1f G rec/s" % (d["workload"], d["kernel_ms"], d["records_per_s"]/1e9))' || exit 1
  echo "# $OTHER, pass $i"
  MGS_LIBRARY=$OTHER timeout -k 10 200 python3 $R/tools/bench_layouts.py $N 7 $W | grep workload | python3 -c 'import sys,json
for l in sys.stdin:
    d=json.loads(l); print("  %-32s %.4f ms  %.1f G rec/s" % (d["workload"], d["kernel_ms"], d["records_per_s"]/1e9))' || exit 1
done
