#!/bin/bash
# Runs on the GPU box: two SQ counter passes (wave-cycle shares, then instruction counts) over a run of any
# tool; prints per-kernel sums.  Usage: bash tools/pmc_sq_tool.sh <tag> <script.py> [args...]
set -e
TAG=$1; shift
SCRIPT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL \
    --output-format csv -d $OUT/pmc1 -- python3 $R/$SCRIPT "$@" > $OUT/out1.json 2> $OUT/log1.txt
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SMEM SQ_BUSY_CYCLES \
    --output-format csv -d $OUT/pmc2 -- python3 $R/$SCRIPT "$@" > $OUT/out2.json 2> $OUT/log2.txt
python3 - $OUT <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            calls[k] += 1
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:4]:
    w = c.get("SQ_WAVE_CYCLES", 1)
    print(k, "launches", calls[k])
    for name, v in sorted(c.items()):
        print(f"   {name:26s} {v:18.0f}  {v / w:8.4f} per wave cycle   {v / max(calls[k], 1):16.0f} per launch")
PY
