#!/bin/bash
# round 6, visit 9 (same script as visit 8, the paired kernel now built from the .inc split): (a) gap_probe (built on the box); (b) the paired-tile build behind its launch rule: parity, then round 5's library ("pre") vs main over paired and unpaired launches
export TMPDIR=/tmp
mkdir -p gpurun_out/r06 tools/probes/bin
hipcc --offload-arch=gfx950 -O2 -w tools/probes/gap_probe.hip -o tools/probes/bin/gap_probe && timeout 600 tools/probes/bin/gap_probe > gpurun_out/r06/v8_gap_probe.txt 2>&1; echo "probe exit $?"; cat gpurun_out/r06/v8_gap_probe.txt
timeout 900 python -m pytest tests/test_m16_gpu.py tests/test_fwd_gpu.py -x -q -k "paired or causal" > gpurun_out/r06/v8_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r06/v8_pytest.log
timeout 1800 python tools/gpu_ab.py --case cfg2,cfg3,cross,dropout,key_bias,causal2k,causal4k,causal,d320_causal4k,d320_causal,causal12k,d1024_causal --rounds 7 --reps 8 pre main > gpurun_out/r06/v8_ab.txt 2>&1; echo "ab exit $?"; grep -E "^(AB)" gpurun_out/r06/v8_ab.txt
