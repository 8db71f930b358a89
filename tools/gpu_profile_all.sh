for w in cfg2 cfg3 cfg4_mask; do PMC_WORKLOAD=$w bash tools/gpu_round.sh wprof > gpurun_out/wprof_$w.log 2>&1; tail -3 gpurun_out/wprof_$w.log | head -2; done
PMC_WORKLOAD=decode PMC_TCC=0 bash tools/gpu_round.sh wprof > gpurun_out/wprof_decode.log 2>&1
bash tools/gpu_round.sh probe > /dev/null 2>&1
QUICK_PYTEST=0 QUICK_WORKLOADS="cfg2 cfg3 cfg4_mask cfg4_offset0 cfg4_nomask cfg2_causal cross gqa attn_mask dropout non_aligned decode" bash tools/gpu_quick.sh
