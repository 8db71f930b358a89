mkdir -p gpurun_out/bench_all
for w in cfg2 cfg3 cfg4_mask cfg4_offset0 cfg4_nomask cfg2_causal cross gqa attn_mask dropout non_aligned decode; do
  extra="--no-cpu-baseline"; [ $w = cfg2 ] && extra=""
  timeout 300 python bench.py --workload $w --steps 20 --warmup 5 $extra > gpurun_out/bench_all/$w.json 2> gpurun_out/bench_all/$w.err; echo "$w exit $?"; tail -2 gpurun_out/bench_all/$w.err | cut -c1-300
  python - <<PY
import json
try:
  d=json.loads(open('gpurun_out/bench_all/$w.json').read().strip().splitlines()[-1])
  print('  ', d['value'], 'TF ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'err', d.get('max_abs_err_vs_sdpa'), d.get('mean_abs_err_vs_sdpa'), 'lse', d.get('max_abs_lse_err'), 'sdpa', d.get('sdpa_gpu_tflops'), 'refproto', d.get('ref_protocol',{}).get('tflops'), d.get('sdpa_error'))
except Exception as e: print('  parse fail', e)
PY
done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --gather > gpurun_out/bench_all/cfg5_dist1.json 2> gpurun_out/bench_all/cfg5_dist1.err; echo "cfg5 dist1 exit $?"; tail -3 gpurun_out/bench_all/cfg5_dist1.err | cut -c1-300; cat gpurun_out/bench_all/cfg5_dist1.json | cut -c1-600
timeout 300 python -m pytest tests/test_sharding_gpu.py -m gpu -x -q 2>&1 | tail -3
