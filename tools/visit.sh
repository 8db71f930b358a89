#!/bin/bash
mkdir -p gpurun_out/v25
timeout 600 python -m pytest tests/test_bench_gpu.py -m gpu -q > gpurun_out/v25/pytest_bench.txt 2>&1; tail -5 gpurun_out/v25/pytest_bench.txt
for w in cross cfg4_mask decode; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/v25/bench_$w.json 2> gpurun_out/v25/bench_$w.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/v25/bench_$w.json').read().strip().splitlines()[-1])
print('$w', d['value'], d['ms_per_step'], d['roofline']['frac'], d['steady_state'])
PY
done
