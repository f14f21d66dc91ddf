#!/bin/bash
# Runs on the GPU box: the whole GPU suite, smoke(), the default bench line, its kernel stats and HBM traffic
# (rocprofv3, separate PMC passes), the wordcount chain.  Stops at the first failure.
set -u
out=gpurun_out/r2
mkdir -p $out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/gputest.log 2>&1
rc=$?
echo rc=$rc >> $out/gputest.log
tail -n 3 $out/gputest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { tail -n 20 $out/smoke.log; exit 1; }
tail -n 1 $out/smoke.log
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err || { tail -n 20 $out/bench.err; exit 1; }
cat $out/bench.json
bash tools/profile_bench.sh r2 > $out/profile_bench.txt 2>&1 || { tail -n 20 $out/profile_bench.txt; exit 1; }
tail -n 30 $out/profile_bench.txt
timeout -k 10 300 python tools/bench_chain.py > $out/chain.json 2> $out/chain.err || { tail -n 20 $out/chain.err; exit 1; }
cat $out/chain.json
