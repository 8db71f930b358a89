#!/bin/bash
# round 6, second evidence visit: the library with the packed-sequence entry points (ffpa_attn_varlen_fwd; the dense kernels' objects are byte-identical to the
# first visit's: tools/visits/gpu_r06_final.sh) — GPU suite, PMC + kernel-trace passes of the profiled workloads (+ varlen), the bench line of every workload
# (+ varlen), the sweep, randomized parity with fresh seeds, smoke
export TMPDIR=/tmp
sha256sum ffpa_attn_amd/libffpa_attn_hip.so | cut -c1-16
FFPA_ROUND=r06 bash tools/gpu_evidence.sh "pytest wprof bench sweep" "cfg2 cfg3 cfg4_mask cfg2_causal attn_mask dropout decode varlen"
O=gpurun_out/final; mkdir -p $O
FFPA_FUZZ_SEEDS=20000:22500 timeout 900 python -m pytest tests/test_fwd_gpu.py -m gpu -q -k test_randomized_against_oracle 2>&1 | tail -2 | tee $O/fuzz_a.txt
FFPA_FUZZ_FLAGS=0x8000 FFPA_FUZZ_SEEDS=22500:25000 timeout 900 python -m pytest tests/test_fwd_gpu.py -m gpu -q -k test_randomized_against_oracle 2>&1 | tail -2 | tee $O/fuzz_b.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
