#!/bin/bash
# A/B of a build whose k_onesweepr workgroups have another size (libmgs_exphalf.so: OSR_THREADS, kOvThreads, wg_per_cu and the image sizes changed by hand) on the byte-granular shapes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in "" $R/mimir_amd/lib/libmgs_exphalf.so "" $R/mimir_amd/lib/libmgs_exphalf.so; do
  echo "== ${lib:-in-tree}"
  for w in words join "octree --level 12"; do
    MGS_LIBRARY=$lib timeout -k 10 200 python $R/bench.py --workload $w --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:30], r['kernel'], 'kernel_ms %.3f'%r['kernel_ms'], 'frac %.3f'%r['frac'])
" || echo "failed: $w"
  done
done
