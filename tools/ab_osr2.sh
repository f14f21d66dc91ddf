# experiments with k_onesweepr's instantiations (runs on the GPU box)
run() {
  echo "== $W $*"
  env "$@" timeout -k 10 200 python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(r['kernel'], 'value %.1f G/s'%(d['value']/1e9), 'kernel_ms %.3f'%r['kernel_ms'], 'frac %.3f'%r['frac'], 'rounds', d['config']['rounds'])
"
}
W=words; run MGS_OSR=0; run MGS_OSR_K=4; run MGS_OSR_K=3 MGS_OSR_PRE=1; run MGS_OSR_K=3 MGS_OSR_PRE=0
W=join; run MGS_OSR=0; run MGS_OSR_K=4;  run MGS_OSR_K=3 MGS_OSR_PRE=1
W="octree --level 15"; run MGS_OSR=0; run MGS_OSR_K=4; run MGS_OSR_K=3
W="octree --level 12"; run MGS_OSR=0; run MGS_OSR_K=4; run MGS_OSR_K=3
W="octree --level 4"; run MGS_OSR=0; run MGS_OSR=1
echo "== phases"
MGS_OSR_STATS=1 MGS_OSR_K=4 timeout -k 10 300 python tools/bench_layouts.py 2e7 3 words,words6,join,oct12,oct4 2>&1 | grep -v amdgpu.ids
MGS_OSR_STATS=1 MGS_OSR_K=3 MGS_OSR_PRE=1 timeout -k 10 300 python tools/bench_layouts.py 2e7 3 words6,join 2>&1 | grep -v amdgpu.ids
