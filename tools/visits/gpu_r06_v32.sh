#!/bin/bash
# round 6, visit 32: the randomized parity tests with the plan pricing a 128- / 304-CU part (FFPA_HIP_FAKE_CUS: other split rules fire — under-filled / one-round / tile ranges at other sizes;
# any plan is a correct launch on any device), final library
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
sha256sum ffpa_attn_amd/libffpa_attn_hip.so | cut -c1-16 | tee gpurun_out/r06/v32_fuzz_fake_cus.txt
for cus in 128 304 64; do
FFPA_HIP_FAKE_CUS=$cus FFPA_FUZZ_SEEDS=50000:51500 timeout 900 python -m pytest tests/test_fwd_gpu.py -m gpu -q -k test_randomized_against_oracle 2>&1 | grep -E "^E  .*Error|passed|failed" | cut -c1-500 | head -4 | sed "s/^/dense, FAKE_CUS=$cus, seeds 50000:51500: /" | tee -a gpurun_out/r06/v32_fuzz_fake_cus.txt
FFPA_HIP_FAKE_CUS=$cus FFPA_VARLEN_FUZZ=4000:4200 timeout 900 python -m pytest tests/test_varlen_gpu.py -m gpu -q -k "test_randomized_packed_batches or test_kv_splits_of_under_filled or test_kv_splits_inside" 2>&1 | grep -E "^E  .*Error|passed|failed" | cut -c1-500 | head -4 | sed "s/^/packed, FAKE_CUS=$cus: /" | tee -a gpurun_out/r06/v32_fuzz_fake_cus.txt
done
FFPA_HIP_FAKE_CUS=64 timeout 900 python -m pytest tests/test_m16_gpu.py -m gpu -q -k "causal_kv_ranges or head_chunk or paired" 2>&1 | grep -E "^E  .*Error|passed|failed" | cut -c1-500 | head -4 | sed "s/^/m16 range tests, FAKE_CUS=64: /" | tee -a gpurun_out/r06/v32_fuzz_fake_cus.txt
