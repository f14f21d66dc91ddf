#!/bin/bash
# Bench lines of the byte-granular shapes for builds of the library with a part of k_onesweepr cut out
# (libmgs_exp<N>.so next to the library; wrong results, times only).  Runs on the GPU box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in "" $R/mimir_amd/lib/libmgs_exp*.so; do
  echo "== ${lib:-in-tree}"
  for w in words join "octree --level 12"; do
    MGS_LIBRARY=$lib timeout -k 10 200 python $R/bench.py --workload $w --no-cpu-baseline --no-verify --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:30], r['kernel'], 'kernel_ms %.3f'%r['kernel_ms'], 'frac %.3f'%r['frac'])
" || echo "failed: $w"
  done
done
