# bench.py over the byte-granular record shapes (runs on the GPU box); extra environment as arguments
for w in words join "octree --level 12" "octree --level 15" "octree --level 4"; do
  env "$@" python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:44], r['kernel'], '%.1f G/s'%(d['value']/1e9), 'kernel_ms %.3f'%r['kernel_ms'], 'frac %.3f'%r['frac'])
"
done
