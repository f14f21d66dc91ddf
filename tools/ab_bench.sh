#!/bin/bash
# A/B of two builds of the library on the headline configuration, alternating, three passes:
# tools/ab_bench.sh <other.so>
R=$(cd "$(dirname "$0")/.." && pwd)
for i in 1 2 3; do
  for lib in "" "$1"; do
    MGS_LIBRARY=$lib timeout -k 10 200 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("  %-16s kernel %.4f ms  step %.4f ms  frac %.4f" % (sys.argv[1] or "in-tree", d["roofline"]["kernel_ms"], d["ms_per_step"], d["roofline"]["frac"]))' "$lib" || exit 1
  done
done
