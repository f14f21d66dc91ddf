#!/bin/bash
# Runs on the GPU box: HBM traffic (FETCH_SIZE and WRITE_SIZE, two passes: they
# cannot share one) of every kernel of a tool run.
# Usage: bash tools/pmc_traffic.sh <tag> <script.py> [args...]   (script relative to the repo root)
set -e
TAG=$1; shift
SCRIPT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/traffic_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- \
    python3 $R/$SCRIPT "$@" > /dev/null 2> $OUT/pmc_fetch.log
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- \
    python3 $R/$SCRIPT "$@" > /dev/null 2> $OUT/pmc_write.log
python3 $R/tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.json
python3 - $OUT/pmc_summary.json <<'PY'
import json, sys
s = json.load(open(sys.argv[1]))
for k, c in sorted(s.items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0)))[:12]:
    f, w = c.get("FETCH_SIZE", 0), c.get("WRITE_SIZE", 0)
    print(f"{k[:64]:64s} read {2 * f * 1024 / 1e6:10.1f} MB  written {w * 1024 / 1e6:10.1f} MB  ({c['dispatches']} launches)")
PY
