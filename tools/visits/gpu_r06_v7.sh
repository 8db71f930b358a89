#!/bin/bash
# round 6, visit 7: (a) gap_probe — how many fillers does one wave per SIMD hide behind its own 16-cycle MFMA?  (b) the kernels with the pass loop of the paired
# tiles around their body vs round 5's library ("pre": 267de97f) on launches that do NOT pair — the loop must cost nothing
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 600 tools/probes/bin/gap_probe > gpurun_out/r06/v7_gap_probe.txt 2>&1; echo "probe exit $?"; cat gpurun_out/r06/v7_gap_probe.txt
timeout 1500 python tools/gpu_ab.py --case cfg2,cfg3,cfg4_mask,cfg4_nomask,cross,gqa,d320,dropout,key_bias,causal,d1024_causal --rounds 7 --reps 8 pre main > gpurun_out/r06/v7_ab.txt 2>&1; echo "ab exit $?"; grep -E "^(AB)" gpurun_out/r06/v7_ab.txt
