mkdir -p gpurun_out/r2
timeout -k 10 300 python -u -m pytest tests/test_gpu_streams.py tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r2/streams_test.log 2>&1; echo rc=$? >> gpurun_out/r2/streams_test.log; tail -n 3 gpurun_out/r2/streams_test.log
timeout -k 10 200 python tools/finder_stats.py > gpurun_out/r2/finder_stats.jsonl 2> gpurun_out/r2/finder_stats.err; tail -n 1 gpurun_out/r2/finder_stats.jsonl
timeout -k 10 200 python tools/finder_stats.py 1e9 16777216 > gpurun_out/r2/finder_stats_16m.jsonl 2>> gpurun_out/r2/finder_stats.err; tail -n 1 gpurun_out/r2/finder_stats_16m.jsonl
for w in words join; do
timeout -k 10 500 python bench.py --workload $w --records 1e8 --chunks 5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2/bench_${w}_5e8.json 2> gpurun_out/r2/bench_${w}_5e8.err || { echo FAIL $w; tail -n 5 gpurun_out/r2/bench_${w}_5e8.err; }
python -c "
import json; d=json.load(open('gpurun_out/r2/bench_${w}_5e8.json')); print('$w', d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['rounds'], d['config']['records_per_gpu'])"
done
