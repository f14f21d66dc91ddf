mkdir -p gpurun_out/r2
timeout -k 10 400 python -u -m pytest tests/test_gpu_unpack.py tests/test_gpu_streams.py tests/test_gpu_cpp_shim.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/unpack_test.log 2>&1; echo rc=$? >> gpurun_out/r2/unpack_test.log; tail -n 4 gpurun_out/r2/unpack_test.log
timeout -k 10 300 python tools/bench_unpack.py > gpurun_out/r2/unpack.jsonl 2> gpurun_out/r2/unpack.err; cat gpurun_out/r2/unpack.jsonl
bash tools/pmc_traffic.sh r2_unpack tools/bench_unpack.py 6e7 2 > gpurun_out/r2/traffic_unpack.txt 2>&1; cat gpurun_out/r2/traffic_unpack.txt | tail -n 8
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2/unpack_stats -- python3 $GRAFT_REPO_ROOT/tools/bench_unpack.py > /dev/null 2>&1; cat $GRAFT_REPO_ROOT/gpurun_out/r2/unpack_stats/*/*kernel_stats.csv | head -8
