#!/bin/bash
# Instantiations of k_onesweepr pinned by the environment (a build with extra instantiations: libmgs_expk.so).
R=${GRAFT_REPO_ROOT:-$(pwd)}
export MGS_LIBRARY=$R/mimir_amd/lib/libmgs_expk.so
run() { local label=$1; shift
  timeout -k 10 200 python $R/bench.py --workload "$@" --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$label', r['kernel'], 'kernel_ms %.3f'%r['kernel_ms'], 'frac %.3f'%r['frac'])
" || echo "failed: $label"
}
for rep in 1 2; do
run "octree12 default (4,2,early)" octree --level 12
MGS_OSR_K=4 MGS_OSR_NQ=3 run "octree12 quad+dword K=4" octree --level 12
MGS_OSR_K=5 MGS_OSR_NQ=3 run "octree12 quad+dword K=5" octree --level 12
done
