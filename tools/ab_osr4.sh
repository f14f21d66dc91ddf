# words through the register kernel that holds a quad and a dword of a record (runs on the GPU box)
run() { echo "== $*"; env "$@" python bench.py --workload words --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(r['kernel'], '%.1f G/s'%(d['value']/1e9), 'kernel_ms %.3f'%r['kernel_ms'], 'frac %.3f'%r['frac'])
"; }
run MGS_OSR=0
run MGS_OSR_FORCE=1 MGS_OSR_K=4 MGS_OSR_NQ=2
run MGS_OSR_FORCE=1 MGS_OSR_K=4 MGS_OSR_NQ=3 MGS_OSR_PRE=1
run MGS_OSR_FORCE=1 MGS_OSR_K=4 MGS_OSR_NQ=3 MGS_OSR_PRE=0
