#!/bin/bash
# Early / late requests of k_onesweepr pinned by the environment on the string-key shapes (in-tree library).
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { local label=$1; shift
  timeout -k 10 200 python $R/bench.py --workload "$@" --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$label', r['kernel'], 'kernel_ms %.3f'%r['kernel_ms'], 'frac %.3f'%r['frac'])
" || echo "failed: $label"
}
for rep in 1 2; do
MGS_OSR_PRE=0 run "words late" words
MGS_OSR_PRE=1 run "words early" words
MGS_OSR_PRE=0 run "join K=4 late" join
MGS_OSR_K=3 MGS_OSR_NQ=2 MGS_OSR_PRE=1 run "join K=3 early" join
MGS_OSR_K=3 MGS_OSR_NQ=2 MGS_OSR_PRE=0 run "join K=3 late" join
done
