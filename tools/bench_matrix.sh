#!/bin/bash
# Runs bench.py over workloads x comm sizes (and the legacy byte-granular kernel) on one GPU;
# one line per run: workload comm legacy G_records/s ms_per_step kernel_ms roofline.frac rounds
# Usage: tools/bench_matrix.sh <out-prefix> [workloads...]
out=${1:-gpurun_out/matrix}; shift
wl=${@:-"u64 words join octree"}
for w in $wl; do
  for c in 0 64M; do
    for legacy in 0 1; do
      if [ "$w" = "u64" ] && [ "$legacy" = "1" ]; then continue; fi
      if [ "$c" != "0" ] && [ "$legacy" = "1" ]; then continue; fi
      f=${out}_${w}_${c}_l${legacy}.json
      MGS_OSV_LEGACY=$legacy timeout -k 10 200 python bench.py --steps 5 --warmup 2 --workload $w --comm-size $c --no-cpu-baseline > $f 2> ${f%.json}.err || { echo "FAIL $w $c $legacy"; tail -3 ${f%.json}.err; continue; }
      python - "$f" "$w" "$c" "$legacy" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], "legacy" if sys.argv[4]=="1" else "new", round(d["value"]/1e9,2), round(d["ms_per_step"],3), round(d["roofline"]["kernel_ms"],3), round(d["roofline"]["frac"],3), d["config"]["rounds"])
PY
    done
  done
done
