#!/bin/bash
# One evidence visit: the GPU suite, the bench line of every workload, the sweep, rocprofv3 kernel trace + PMC passes for the profiled
# workloads — everything under gpurun_out/ in the layout tools/collect_profiles.sh copies into profiles/.
#   usage: FFPA_GIT_HEAD=<sha> [FFPA_ROUND=r05] bash tools/gpu_evidence.sh ["pytest wprof bench sweep"] ["cfg2 cfg3 ..."]
STAGES=${1:-"pytest wprof bench sweep"}  # (wprof first: the bench lines then quote roofline.traffic from the PMC summaries of this very visit)
ROUND=${FFPA_ROUND:-r04}
WPROF=${2:-"cfg2 cfg3 cfg4_mask attn_mask dropout decode"}
export TMPDIR=/tmp
mkdir -p gpurun_out/final
for st in $STAGES; do
case $st in
pytest)
  timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/final/pytest.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/final/pytest.log ;;
bench)
  for w in cfg2 cfg3 cfg4_mask cfg4_offset0 cfg4_nomask cfg2_causal cross gqa gqa_causal prompt_tp8 attn_mask dropout non_aligned decode varlen varlen_decode; do
    extra="--no-cpu-baseline"; [ $w = cfg2 ] && extra=""
    timeout 300 python bench.py --workload $w --steps 20 --warmup 5 $extra > gpurun_out/final/bench_$w.json 2> gpurun_out/final/bench_$w.err
    python - <<PY
import json
try:
  d=json.loads(open('gpurun_out/final/bench_$w.json').read().strip().splitlines()[-1])
  dv=d.get('device',{})
  ss=d.get('steady_state') or {}
  print('BENCH %-13s %8.2f TF %8.4f ms frac %.4f @clk %s MHz (rate frac %s) | steady %s TF @ %s MHz %s W | err %s/%s sdpa %s' % ('$w', d['value'], d['ms_per_step'], d['roofline']['frac'], dv.get('sclk_mhz_avg'), d['roofline'].get('frac_of_mfma_rate_at_measured_clock'), ss.get('tflops'), ss.get('sclk_mhz_avg'), ss.get('power_w_avg'), d.get('max_abs_err_vs_sdpa'), d.get('max_abs_err_vs_fp32_math'), d.get('sdpa_gpu_tflops')))
except Exception as e: print('BENCH $w parse fail', e)
PY
  done
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final/bench_cfg5_1gpu.json 2> gpurun_out/final/bench_cfg5_1gpu.err; echo "cfg5 exit $?" ;;
sweep)
  timeout 600 python bench.py --sweep --steps 10 > gpurun_out/final/sweep.json 2> gpurun_out/final/sweep.txt; echo "sweep exit $?"; grep -v amdgpu.ids gpurun_out/final/sweep.txt | tail -32 ;;
wprof)
  for w in $WPROF; do
    tcc=1; [ $w = decode ] && tcc=0
    PMC_WORKLOAD=$w PMC_TCC=$tcc bash tools/gpu_round.sh wprof > gpurun_out/wprof_$w.log 2>&1
    grep -E "exit [1-9]" gpurun_out/wprof_$w.log | head -3
    # (on the box only: the snapshot's profiles/ gets this visit's summary, so that the bench stage below finds a profile of the running library)
    [ -s gpurun_out/wprof_$w/summary.json ] && cp gpurun_out/wprof_$w/summary.json profiles/${ROUND}_bench_${w}_pmc.json
    python - <<PY
import json
try:
  d=json.load(open('gpurun_out/wprof_$w/summary.json'))
  k=d['derived']
  print('WPROF %-10s trace median %.4f ms frac %.4f | mfma busy %.3f clock %.2f GHz | hbm x algorithmic %.2f | L2 hit %.3f' % ('$w', k['kernel_trace']['median_ms'], k['kernel_trace'].get('frac_of_peak_at_median', 0), k.get('mfma_busy_fraction_of_simd_cycles', 0), k.get('effective_clock_ghz_in_that_pass', 0), (k.get('hbm_read_bytes_corrected_x2', 0) + k.get('hbm_write_bytes', 0)) / max(1, k.get('algorithmic_bytes_Q+K+V+O+LSE', 1)), k.get('l2_hit_rate', 0)))
except Exception as e: print('WPROF $w parse fail', e)
PY
  done ;;
esac
done
