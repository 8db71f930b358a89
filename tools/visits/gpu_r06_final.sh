#!/bin/bash
# round 6, the evidence visit of the final library: GPU suite, PMC + kernel-trace passes of the profiled workloads, the bench line of every workload (roofline.traffic measured in each run),
# the sweep, and the randomized parity runs with fresh seeds (default plan; paired row tiles forced)
export TMPDIR=/tmp
sha256sum ffpa_attn_amd/libffpa_attn_hip.so | cut -c1-16
FFPA_ROUND=r06 bash tools/gpu_evidence.sh "pytest wprof bench sweep" "cfg2 cfg3 cfg4_mask cfg2_causal attn_mask dropout decode"
O=gpurun_out/final; mkdir -p $O
FFPA_FUZZ_SEEDS=10000:12500 timeout 900 python -m pytest tests/test_fwd_gpu.py -m gpu -q -k test_randomized_against_oracle 2>&1 | tail -2 | tee $O/fuzz_a.txt
FFPA_FUZZ_FLAGS=0x8000 FFPA_FUZZ_SEEDS=12500:15000 timeout 900 python -m pytest tests/test_fwd_gpu.py -m gpu -q -k test_randomized_against_oracle 2>&1 | tail -2 | tee $O/fuzz_b.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
