#!/bin/bash
# Records per lane of k_onesweepr pinned by the environment (a build with the extra instantiations: libmgs_expk.so).
R=${GRAFT_REPO_ROOT:-$(pwd)}
export MGS_LIBRARY=$R/mimir_amd/lib/libmgs_expk.so
run() { # label, workload args...
  local label=$1; shift
  timeout -k 10 200 python $R/bench.py --workload "$@" --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$label', d['config']['workload'][:24], r['kernel'], 'kernel_ms %.3f'%r['kernel_ms'], 'frac %.3f'%r['frac'])
" || echo "failed: $label"
}
for rep in 1 2; do
run "join default" join
MGS_OSR_K=4 MGS_OSR_NQ=2 MGS_OSR_PRE=1 run "join K=4 pre" join
run "octree15 default" octree --level 15
MGS_OSR_K=4 MGS_OSR_NQ=2 MGS_OSR_PRE=1 run "octree15 K=4 pre" octree --level 15
MGS_OSR_K=4 MGS_OSR_NQ=2 MGS_OSR_PRE=0 run "octree15 K=4" octree --level 15
run "octree4 default" octree --level 4
MGS_OSR_K=5 MGS_OSR_NQ=1 MGS_OSR_PRE=1 run "octree4 K=5" octree --level 4
MGS_OSR_K=6 MGS_OSR_NQ=1 MGS_OSR_PRE=1 run "octree4 K=6" octree --level 4
run "octree12 default" octree --level 12
MGS_OSR_K=4 MGS_OSR_NQ=2 MGS_OSR_PRE=1 run "octree12 K=4 pre" octree --level 12
done
