#!/bin/bash
# quick regression + mask check on one MI355X: pytest (optional), a few bench workloads (value / ms / frac / errors)
mkdir -p gpurun_out/quick
[ "${QUICK_PYTEST:-1}" = 1 ] && (timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4)
for w in ${QUICK_WORKLOADS:-cfg2 cfg3 cfg4_mask cfg4_offset0}; do
  timeout 300 python bench.py --workload $w --steps ${QUICK_STEPS:-30} --warmup 5 --no-cpu-baseline > gpurun_out/quick/$w.json 2> gpurun_out/quick/$w.err || tail -3 gpurun_out/quick/$w.err
  python - <<PY
import json
try:
  d=json.loads(open('gpurun_out/quick/$w.json').read().strip().splitlines()[-1])
  print('QUICK %-13s %8.2f TF  %8.4f ms  frac %.4f  kernel_ms %.4f  err %s mean %s lse %s sdpa %s'%('$w', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d.get('max_abs_err_vs_sdpa'), d.get('mean_abs_err_vs_sdpa'), d.get('max_abs_lse_err'), d.get('sdpa_gpu_tflops')))
except Exception as e: print('QUICK $w parse fail', e)
PY
done
