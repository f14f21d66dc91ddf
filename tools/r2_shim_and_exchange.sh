mkdir -p gpurun_out/r2
timeout -k 10 400 python -u -m pytest tests/test_gpu_cpp_shim.py tests/test_gpu_shuffle.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/shim_test.log 2>&1; echo rc=$? >> gpurun_out/r2/shim_test.log; tail -n 4 gpurun_out/r2/shim_test.log
timeout -k 10 600 python tools/bench_shim_write.py 1.2e7 > gpurun_out/r2/shim_write.jsonl 2> gpurun_out/r2/shim_write.err; cat gpurun_out/r2/shim_write.jsonl | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['record_shape'], {k:(round(v['ns_per_record'],2)) for k,v in d.items() if isinstance(v,dict)})"
for ex in batch rounds; do
MGS_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --records 2e7 --steps 2 --warmup 1 --exchange $ex > gpurun_out/r2/bench_gloo2_$ex.json 2> gpurun_out/r2/bench_gloo2_$ex.err || { echo "FAIL gloo2 $ex"; tail -n 5 gpurun_out/r2/bench_gloo2_$ex.err; }
python -c "
import json; d=json.load(open('gpurun_out/r2/bench_gloo2_$ex.json')); print('gloo2 $ex', d['n_gpus'], d['value'], d['config']['rounds'], d['exchange'])"
done
