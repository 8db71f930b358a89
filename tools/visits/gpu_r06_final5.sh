#!/bin/bash
# round 6, fifth evidence visit: the library with the compact grid of ragged packed prefill batches on top of the fourth visit's library (KV ranges per row tile in both modes of the packed kernel) — one
# round (ffpa_capi.o and the ffpa_varlen_d*.o objects changed; the 22 dense kernel objects are byte-identical to every earlier visit's) — GPU suite, PMC + kernel-trace passes of
# the profiled workloads (+ prompt_tp8), the bench line of every workload (+ prompt_tp8), the sweep, randomized parity with fresh seeds, smoke
export TMPDIR=/tmp
sha256sum ffpa_attn_amd/libffpa_attn_hip.so | cut -c1-16
FFPA_ROUND=r06 bash tools/gpu_evidence.sh "pytest wprof bench sweep" "cfg2 cfg3 cfg4_mask cfg2_causal attn_mask dropout decode varlen varlen_decode prompt_tp8"
O=gpurun_out/final; mkdir -p $O
FFPA_FUZZ_SEEDS=52000:54500 timeout 900 python -m pytest tests/test_fwd_gpu.py -m gpu -q -k test_randomized_against_oracle 2>&1 | tail -2 | tee $O/fuzz_a.txt
FFPA_FUZZ_FLAGS=0x100040 FFPA_FUZZ_SPLITS=3 FFPA_FUZZ_SEEDS=54500:57000 timeout 900 python -m pytest tests/test_fwd_gpu.py -m gpu -q -k test_randomized_against_oracle 2>&1 | tail -2 | tee $O/fuzz_b.txt
FFPA_VARLEN_FUZZ=5000:5300 timeout 900 python -m pytest tests/test_varlen_gpu.py -m gpu -q -k test_randomized_packed_batches 2>&1 | tail -2 | tee $O/fuzz_c.txt
FFPA_VARLEN_FUZZ_SHORT=5000:5400 timeout 900 python -m pytest tests/test_varlen_gpu.py -m gpu -q -k test_randomized_short_query_batches 2>&1 | tail -2 | tee $O/fuzz_d.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
