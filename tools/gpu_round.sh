#!/bin/bash
# One GPU-box visit: diagnostics ladder, GPU test suite, bench line, rocprofv3 kernel trace.
# Everything lands in gpurun_out/ (merged back by gpurun).  Each stage has its own timeout.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
STAGES=${1:-"diag pytest bench prof"}
for st in $STAGES; do
case $st in
diag)
  timeout 900 python tools/gpu_diag.py > gpurun_out/diag.log 2>&1; echo "diag exit $?" | tee -a gpurun_out/diag.log ;;
pytest)
  timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/pytest.log
  tail -5 gpurun_out/pytest.log ;;
pytestall)
  timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/pytest.log
  tail -30 gpurun_out/pytest.log ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" | tee -a gpurun_out/smoke.log ;;
bench)
  timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"; cat gpurun_out/bench.json ;;
prof)
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof -o trace -- python $OLDPWD/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-sdpa) > gpurun_out/prof.log 2>&1
  echo "prof exit $?"; find gpurun_out/prof -name "*stats*" | head; ;;
pmc)
  rm -rf gpurun_out/pmc
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OLDPWD/gpurun_out/pmc -o pmc1 -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sdpa) > gpurun_out/pmc1.log 2>&1
  echo "pmc1 exit $?"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d $OLDPWD/gpurun_out/pmc -o pmc2 -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sdpa) > gpurun_out/pmc2.log 2>&1
  echo "pmc2 exit $?"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $OLDPWD/gpurun_out/pmc -o pmc3 -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sdpa) > gpurun_out/pmc3.log 2>&1
  echo "pmc3 exit $?" ;;
pmcsq)
  rm -rf gpurun_out/pmc
  (cd /tmp && timeout 180 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OLDPWD/gpurun_out/pmc -o pmc1 -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sdpa) > gpurun_out/pmc1.log 2>&1
  echo "pmc1 exit $?"
  (cd /tmp && timeout 180 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU -d $OLDPWD/gpurun_out/pmc -o pmc3 -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sdpa) > gpurun_out/pmc3.log 2>&1
  echo "pmc3 exit $?" ;;
pmcfetch)
  (cd /tmp && timeout 90 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OLDPWD/gpurun_out/pmc -o pmc2 -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sdpa) > gpurun_out/pmc2.log 2>&1
  echo "pmcfetch exit $?" ;;
pmcwrite)
  (cd /tmp && timeout 90 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OLDPWD/gpurun_out/pmc -o pmc4 -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sdpa) > gpurun_out/pmc4w.log 2>&1
  echo "pmcwrite exit $?" ;;
dist1)
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-sdpa > gpurun_out/dist1.json 2> gpurun_out/dist1.err; echo "dist1 exit $?"; cat gpurun_out/dist1.json; tail -3 gpurun_out/dist1.err
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-sdpa --gather > gpurun_out/dist1g.json 2>> gpurun_out/dist1.err; echo "dist1 gather exit $?" ;;
work)
  for w in cfg3 cfg2_causal cfg4; do timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$w.json 2>> gpurun_out/bench.err; echo "$w exit $?"; cat gpurun_out/bench_$w.json; done ;;
pmc4)
  rm -rf gpurun_out/pmc4
  (cd /tmp && timeout 180 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES -d $OLDPWD/gpurun_out/pmc4 -o p -- python $OLDPWD/bench.py --workload ${PMC_WORKLOAD:-cfg4} --steps 3 --warmup 1 --no-cpu-baseline --no-sdpa) > gpurun_out/pmc4.log 2>&1
  echo "pmc4 exit $?" ;;
clocks)
  rm -rf gpurun_out/clk
  (cd /tmp && timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OLDPWD/gpurun_out/clk -o clk -- python $OLDPWD/tools/gpu_ab.py $CLK_ARGS --rounds 2 --reps 3) > gpurun_out/clk.log 2>&1
  echo "clocks exit $?"; grep "^AB" gpurun_out/clk.log ;;
pmcab)
  rm -rf gpurun_out/pmcab
  (cd /tmp && timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc ${PMCAB_COUNTERS:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY} -d $OLDPWD/gpurun_out/pmcab -o ab -- python $OLDPWD/tools/gpu_ab.py $PMCAB_ARGS --rounds 2 --reps 3) > gpurun_out/pmcab.log 2>&1
  echo "pmcab exit $?"; python tools/parse_pmc_ab.py gpurun_out/pmcab/ab_counter_collection.csv 3 $PMCAB_TAGS ;;
biasperf)
  timeout 600 python tools/gpu_bias_bench.py > gpurun_out/bias.log 2>&1; echo "bias exit $?"; grep BIAS gpurun_out/bias.log; tail -2 gpurun_out/bias.log | grep -v BIAS ;;
decode)
  timeout 600 python tools/gpu_decode_bench.py > gpurun_out/decode.log 2>&1; echo "decode exit $?"; grep DECODE gpurun_out/decode.log; tail -3 gpurun_out/decode.log | grep -v DECODE ;;
ab)
  timeout 600 python tools/gpu_ab.py $AB_ARGS > gpurun_out/ab.log 2>&1; echo "ab exit $?"; cat gpurun_out/ab.log ;;
ab2)
  timeout 600 python tools/gpu_ab.py $AB2_ARGS > gpurun_out/ab2.log 2>&1; echo "ab2 exit $?"; cat gpurun_out/ab2.log ;;
wprof)
  # rocprofv3 kernel stats + PMC passes for one bench workload ($PMC_WORKLOAD, default cfg3); counters beyond the SQ sets are
  # one pass each, short timeouts: a TCC pass hung on this pool in round 1, so those run last
  W=${PMC_WORKLOAD:-cfg3}; OUT=gpurun_out/wprof_$W; rm -rf $OUT; mkdir -p $OUT
  BENCH="python $PWD/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-sdpa --no-steady"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT -o trace -- python $OLDPWD/bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-sdpa --no-steady) > $OUT/trace.log 2>&1; echo "wprof $W trace exit $?"
  pass() { n=$1; shift; (cd /tmp && timeout -s KILL ${PASS_TIMEOUT:-120} rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OLDPWD/$OUT -o $n -- $BENCH) > $OUT/$n.log 2>&1; echo "wprof $W $n exit $?"; }
  pass sq1 SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE
  pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_VALU
  pass sq3 SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  pass tcp1 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
  pass ta1 TA_TA_BUSY_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum
  pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
  pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_RFIFO_STALL_CYCLES_sum
  if [ "${PMC_TCC:-1}" = 1 ]; then
    PASS_TIMEOUT=75 pass tcc1 TCC_REQ_sum TCC_READ_sum
    PASS_TIMEOUT=75 pass tcc2 TCC_HIT_sum TCC_MISS_sum
    PASS_TIMEOUT=75 pass tcc3 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum
  fi
  python tools/pmc_summary.py $OUT $OUT/summary.json "workload $W" > $OUT/summary.log 2>&1; tail -40 $OUT/summary.log ;;
probe)
  mkdir -p tools/probes/bin
  [ -x tools/probes/bin/stream_probe ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/stream_probe.hip -o tools/probes/bin/stream_probe
  timeout 300 tools/probes/bin/stream_probe > gpurun_out/stream_probe.log 2>&1; echo "probe exit $?"; cat gpurun_out/stream_probe.log ;;
ab3)
  timeout 600 python tools/gpu_ab.py $AB3_ARGS > gpurun_out/ab3.log 2>&1; echo "ab3 exit $?"; cat gpurun_out/ab3.log ;;
rocminfo)
  (rocminfo | grep -E "Name:|Compute Unit|Max Clock|gfx" | head -40; rocm-smi --showclocks 2>/dev/null | head -30; rocprofv3 -L 2>/dev/null | grep -c "Name") > gpurun_out/rocminfo.log 2>&1 ;;
esac
done
ls -la gpurun_out | head -30
