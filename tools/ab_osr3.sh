# what does the chained scan cost k_onesweepr?  (runs on the GPU box; the fake-chain runs write wrong output)
for w in words6 join oct12; do
  for k in "MGS_OSR_K=4" "MGS_OSR_K=3 MGS_OSR_PRE=1"; do
    for f in 0 1; do
      echo "== $w $k fakechain=$f"
      env $k MGS_OSR_FAKECHAIN=$f MGS_OSV_STATS=2 timeout -k 10 200 python tools/bench_layouts.py 2e7 3 $w 2>&1 | grep workload | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['kernel'], 'kernel_ms %.3f'%d['kernel_ms'], 'frac %.3f'%d['frac_of_8TBps'])
"
    done
  done
done
