#!/bin/bash
# Runs on the GPU box: the parse kernels (unpack, tokenizer) with the stream cut into 4096 / 8192 / 16384
# chunks (one wave each): more waves per SIMD against a longer chain through the chunk tables.
mkdir -p gpurun_out/r2
for c in 2048 3072 4096 2048 3072; do
  echo "chunks $c"
  MGS_PARSE_CHUNKS=$c timeout -k 10 200 python tools/bench_unpack.py 2>/dev/null | head -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('  unpack  ', round(d['ms'],4), 'ms', round(d['records_per_s']/1e9,1), 'G rec/s')"
  MGS_PARSE_CHUNKS=$c timeout -k 10 200 python tools/bench_tokenize.py 2>/dev/null | head -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('  tokenize', d.get('ms'), 'ms', d.get('words_per_s', d.get('records_per_s')))"
done
