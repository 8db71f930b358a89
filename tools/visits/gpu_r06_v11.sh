#!/bin/bash
# round 6, visit 11: the wave-rotated piece schedule of the pipelined split-D loop (variant burst: a phase's piece slots dealt to the four waves in turn, the owner issuing four
# pieces back to back) — bit-identity + speed, per-phase cycles, and the TA / TCP / MFMA counters of both arms on config 3
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python tools/gpu_ab.py --case cfg3,d1024_causal,n2048_d1024,cross_d1024,gqa_d1024 --rounds 7 --reps 6 main burst > gpurun_out/r06/v11_ab.txt 2>&1; echo "ab exit $?"; grep -E "^(AB)" gpurun_out/r06/v11_ab.txt
timeout 300 python tools/gpu_phase_times.py --dims 1024 timing timing_burst > gpurun_out/r06/v11_phase.txt 2>&1; echo "phase exit $?"; grep PHASE gpurun_out/r06/v11_phase.txt
for arm in main burst; do
  OUT=gpurun_out/r06/v11_pmc_$arm; rm -rf $OUT; mkdir -p $OUT
  LIBENV=""; [ $arm = burst ] && LIBENV="FFPA_HIP_LIBRARY=$PWD/ffpa_attn_amd/variants/libffpa_attn_hip_burst.so"
  BENCH="python $PWD/bench.py --workload cfg3 --steps 3 --warmup 1 --no-cpu-baseline --no-sdpa --no-steady"
  pass() { n=$1; shift; (cd /tmp && env $LIBENV timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OLDPWD/$OUT -o $n -- $BENCH) > $OUT/$n.log 2>&1; echo "pmc $arm $n exit $?"; }
  pass sq1 SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE
  pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_VALU
  pass ta1 TA_TA_BUSY_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum
  pass tcp1 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
  pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_RFIFO_STALL_CYCLES_sum
  python tools/pmc_summary.py $OUT $OUT/summary.json "workload cfg3 arm $arm" > $OUT/summary.log 2>&1
  python - <<PY
import json
try:
  d=json.load(open('$OUT/summary.json')); k=d['derived']
  ta=d.get('TA_TA_BUSY_sum',{}).get('per_dispatch_mean',0); cyc=k.get('cycles_per_xcd',1)
  pend=d.get('TCP_PENDING_STALL_CYCLES_sum',{}).get('per_dispatch_mean',0)
  lat=d.get('TCP_TCC_READ_REQ_LATENCY_sum',{}).get('per_dispatch_mean',0); req=d.get('TCP_TCC_READ_REQ_sum',{}).get('per_dispatch_mean',1)
  print('PMC %-6s kernel %.3f ms | mfma busy %.3f | TA busy %.3f | sum %.3f | TCP pending stall %.3f | L2 read latency %.0f cycles, %.1f lines in flight per CU | wait_inst_any %.3f' % ('$arm', k.get('kernel_ms_mean_under_pmc',0), k.get('mfma_busy_fraction_of_simd_cycles',0), ta/256/cyc, k.get('mfma_busy_fraction_of_simd_cycles',0)+ta/256/cyc, pend/256/cyc, lat/max(req,1), lat/256/cyc, k.get('wait_inst_any_frac',0)))
except Exception as e: print('PMC $arm parse fail', e)
PY
done
