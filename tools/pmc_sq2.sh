#!/bin/bash
# Runs on the GPU box: two SQ counter passes over a short bench run (wave cycle shares, then
# instruction counts); prints the per-kernel sums.  Usage: bash tools/pmc_sq2.sh <tag> <bench args...>
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL \
    --output-format csv -d $OUT/pmc1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench1.json 2> $OUT/log1.txt
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SMEM SQ_BUSY_CYCLES \
    --output-format csv -d $OUT/pmc2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench2.json 2> $OUT/log2.txt
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES",):
            calls[k] += 1
try:
    n = json.load(open(sys.argv[1] + "/bench1.json"))["config"]["records_per_gpu"]
except Exception:
    n = None
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:3]:
    w = c.get("SQ_WAVE_CYCLES", 1)
    print(k, "calls", calls[k], "records/call", n)
    for name, v in sorted(c.items()):
        per = f"{v / calls[k] / n:9.2f} per record" if n and calls[k] and name.startswith("SQ_INSTS") else ""
        print(f"   {name:26s} {v:16.0f}  {v / w:6.3f} of wave cycles {per}")
PY
