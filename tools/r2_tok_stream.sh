#!/bin/bash
# Streaming tokenizer (k_tok_count / k_tok_scan / k_tok_stream): parity tests, then the call timed at several
# chunk counts (MGS_TOK_CHUNKS), kernel stats and HBM traffic.  Stops at the first failing GPU step.
set -u
out=gpurun_out/r2
mkdir -p $out
timeout -k 10 300 python -u -m pytest tests/test_gpu_tokenize.py tests/test_gpu_wordcount_chain.py -x -q \
    --timeout 120 --timeout-method thread > $out/tok_stream_test.log 2>&1
rc=$?
echo rc=$rc >> $out/tok_stream_test.log
tail -n 15 $out/tok_stream_test.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/bench_tokenize.py > $out/tokenize.jsonl 2> $out/tok_one.err || exit 1
cat $out/tokenize.jsonl
: > $out/tok_chunks.jsonl
for c in 2048 4096 8192 16384; do
    MGS_TOK_CHUNKS=$c timeout -k 10 200 python tools/bench_tokenize.py > $out/tok_one.json 2> $out/tok_one.err || exit 1
    echo "{\"chunks\": $c, \"line\": $(cat $out/tok_one.json)}" >> $out/tok_chunks.jsonl
done
python - <<'P'
import json
for l in open('gpurun_out/r2/tok_chunks.jsonl'):
    d = json.loads(l); print(d['chunks'], round(d['line']['ms'], 4), round(d['line']['frac_of_8TBps'], 3))
P
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv \
    -d $GRAFT_REPO_ROOT/gpurun_out/r2/tok_stream_stats -- python3 $GRAFT_REPO_ROOT/tools/bench_tokenize.py > /dev/null 2>&1) || exit 1
cat gpurun_out/r2/tok_stream_stats/*/*kernel_stats.csv | head -8
bash tools/pmc_traffic.sh r2_tokenize tools/bench_tokenize.py 6e7 2 > $out/traffic_tokenize.txt 2>&1 || exit 1
tail -n 6 $out/traffic_tokenize.txt
