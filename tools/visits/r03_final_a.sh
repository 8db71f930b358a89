#!/bin/bash
# round 3 evidence, part A (one MI355X): GPU test suite, smoke, every bench workload, the one-shot sweep, the sharded path at world size 1
mkdir -p gpurun_out/final
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/final/pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/final/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/final/smoke.log
for w in cfg2 cfg3 cfg4_mask cfg4_offset0 cfg4_nomask cfg2_causal cross gqa attn_mask dropout non_aligned decode; do
  extra="--no-cpu-baseline"; [ $w = cfg2 ] && extra=""
  timeout 400 python bench.py --workload $w --steps 20 --warmup 5 $extra > gpurun_out/final/bench_$w.json 2> gpurun_out/final/bench_$w.err; echo "$w exit $?"; tail -1 gpurun_out/final/bench_$w.err | cut -c1-200
  python - <<PY
import json
try:
  d=json.loads(open('gpurun_out/final/bench_$w.json').read().strip().splitlines()[-1])
  print('   %-13s %8.2f  ms %8.4f  frac %.4f  kernel %s | err sdpa %s math %s (sdpa vs math %s) lse %s | sdpa %s TF' % ('$w', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d.get('max_abs_err_vs_sdpa'), d.get('max_abs_err_vs_fp32_math'), d.get('sdpa_max_abs_err_vs_fp32_math'), d.get('max_abs_lse_err'), d.get('sdpa_gpu_tflops')))
except Exception as e: print('   parse fail', e)
PY
done
timeout 600 python bench.py --sweep --steps 10 > gpurun_out/final/sweep.json 2> gpurun_out/final/sweep.txt; echo "sweep exit $?"; cat gpurun_out/final/sweep.txt | cut -c1-200
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final/bench_cfg5_1gpu.json 2> gpurun_out/final/bench_cfg5_1gpu.err; echo "cfg5 (torchrun, 1 rank) exit $?"; cut -c1-400 gpurun_out/final/bench_cfg5_1gpu.json
timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/final/bench_gpus2.json 2> gpurun_out/final/bench_gpus2.err; echo "self-spawned --gpus 2 on a 1-GPU box: exit $? (expected: non-zero, no hang)"; tail -2 gpurun_out/final/bench_gpus2.err | cut -c1-200
