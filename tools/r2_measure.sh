set -e
mkdir -p gpurun_out/r2
timeout -k 10 300 python bench.py > gpurun_out/r2/bench_default.json 2> gpurun_out/r2/bench_default.err
timeout -k 10 200 python bench.py --comm-size 0 --no-cpu-baseline > gpurun_out/r2/bench_oneround.json 2> gpurun_out/r2/bench_oneround.err
timeout -k 10 200 python bench.py --comm-size 1G --no-cpu-baseline > gpurun_out/r2/bench_1g.json 2> gpurun_out/r2/bench_1g.err
bash tools/bench_matrix.sh gpurun_out/r2/matrix words join octree > gpurun_out/r2/matrix.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2/prof_default -o default -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2/bench_default_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2/rocprof_default.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2/prof_words -o words -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload words --comm-size 0 > $GRAFT_REPO_ROOT/gpurun_out/r2/bench_words_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2/rocprof_words.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2/prof_words64 -o words64 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload words > $GRAFT_REPO_ROOT/gpurun_out/r2/bench_words64_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2/rocprof_words64.err
ls $GRAFT_REPO_ROOT/gpurun_out/r2/prof_default
