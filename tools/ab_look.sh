#!/bin/bash
# Look-back window of k_onesweepr's chained scan: builds with LOOK = 4 and 16 (libmgs_explook<N>.so) against the in-tree 8.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for lib in "" $R/mimir_amd/lib/libmgs_explook*.so; do
  echo "== ${lib:-in-tree}"
  for w in words join "octree --level 12"; do
    MGS_LIBRARY=$lib timeout -k 10 200 python $R/bench.py --workload $w --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:30], 'kernel_ms %.3f'%r['kernel_ms'], 'frac %.3f'%r['frac'])
" || echo "failed: $w"
  done
done
done
