#!/bin/bash
# round 6, third evidence visit: the library with C-ABI 6 (KV splits inside the packed call, (head, token) rows, the NT build; the dense kernels' objects are byte-identical
# to the first two visits': tools/visits/gpu_r06_final.sh / _final2.sh) — GPU suite, PMC + kernel-trace passes of the profiled workloads (+ varlen_decode), the bench line of
# every workload (+ varlen_decode), the sweep, randomized parity with fresh seeds (dense; packed batches; short-query packed batches in every launch form), smoke
export TMPDIR=/tmp
sha256sum ffpa_attn_amd/libffpa_attn_hip.so | cut -c1-16
FFPA_ROUND=r06 bash tools/gpu_evidence.sh "pytest wprof bench sweep" "cfg2 cfg3 cfg4_mask cfg2_causal attn_mask dropout decode varlen varlen_decode"
O=gpurun_out/final; mkdir -p $O
FFPA_FUZZ_SEEDS=30000:32500 timeout 900 python -m pytest tests/test_fwd_gpu.py -m gpu -q -k test_randomized_against_oracle 2>&1 | tail -2 | tee $O/fuzz_a.txt
FFPA_FUZZ_FLAGS=0x8000 FFPA_FUZZ_SEEDS=32500:35000 timeout 900 python -m pytest tests/test_fwd_gpu.py -m gpu -q -k test_randomized_against_oracle 2>&1 | tail -2 | tee $O/fuzz_b.txt
FFPA_VARLEN_FUZZ=1000:1300 timeout 900 python -m pytest tests/test_varlen_gpu.py -m gpu -q -k test_randomized_packed_batches 2>&1 | tail -2 | tee $O/fuzz_c.txt
FFPA_VARLEN_FUZZ_SHORT=1000:1400 timeout 900 python -m pytest tests/test_varlen_gpu.py -m gpu -q -k test_randomized_short_query_batches 2>&1 | tail -2 | tee $O/fuzz_d.txt
timeout 600 python tools/gpu_varlen_splits.py 2>&1 | grep -v amdgpu.ids > $O/varlen_splits.txt; grep -c VARLENSPLITS $O/varlen_splits.txt
timeout 600 python tools/gpu_varlen_decode.py 2>&1 | grep -v amdgpu.ids > $O/varlen_decode.txt; grep -c "^VARLENDECODE " $O/varlen_decode.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
