#!/bin/bash
# round 3 evidence, part B: rocprofv3 kernel trace + PMC passes (tools/gpu_round.sh stage wprof) per workload, summarised with provenance
for w in ${WPROF_WORKLOADS:-cfg2 cfg3 cfg4_mask attn_mask dropout}; do PMC_WORKLOAD=$w bash tools/gpu_round.sh wprof > gpurun_out/wprof_$w.log 2>&1; grep -E "exit|frac_of_peak_at_median|mfma_busy|effective_clock" gpurun_out/wprof_$w.log | tr '\n' ' ' | cut -c1-600; echo; done
PMC_WORKLOAD=decode PMC_TCC=0 bash tools/gpu_round.sh wprof > gpurun_out/wprof_decode.log 2>&1; grep -E "frac_of_peak_at_median" gpurun_out/wprof_decode.log
