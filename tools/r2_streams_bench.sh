#!/bin/bash
# Runs on the GPU box: the bench in the streams layout (default) and in the image layout, u64 and the other shapes.
mkdir -p gpurun_out/r2
for w in u64 words join octree; do
  for lay in streams images; do
    f=gpurun_out/r2/bench_${w}_64M_${lay}.json
    timeout -k 10 200 python bench.py --steps 5 --warmup 2 --workload $w --layout $lay --no-cpu-baseline > $f 2> ${f%.json}.err || { echo "FAIL $w $lay"; tail -n 3 ${f%.json}.err; exit 1; }
    python - "$f" "$w" "$lay" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], round(d["value"]/1e9,2), "Grec/s", round(d["ms_per_step"],3), "ms/step kernel", round(d["roofline"]["kernel_ms"],3), "frac", round(d["roofline"]["frac"],3), "rounds", d["config"]["rounds"], d["roofline"]["kernel"][:40])
PY
  done
done
MGS_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --records 2e7 --steps 2 --warmup 1 > gpurun_out/r2/bench_gloo2_streams.json 2> gpurun_out/r2/bench_gloo2_streams.err || { echo "FAIL gloo2"; tail -n 5 gpurun_out/r2/bench_gloo2_streams.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r2/bench_gloo2_streams.json')); print('gloo2', d['n_gpus'], d['value'], d['config']['rounds'], d['exchange'])"
