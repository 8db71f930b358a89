#!/bin/bash
# round 6, third evidence visit, second part (same library 26fc9a3f...: the PAIR build measured in between was removed again, the rebuilt library is byte-identical): the GPU suite
# and the packed randomized test once more after the oracle check's one-flip allowance (tests only)
export TMPDIR=/tmp
sha256sum ffpa_attn_amd/libffpa_attn_hip.so | cut -c1-16
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/final/pytest.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/final/pytest.log
FFPA_VARLEN_FUZZ=1000:1300 timeout 900 python -m pytest tests/test_varlen_gpu.py -m gpu -q -k test_randomized_packed_batches 2>&1 | tail -2 | tee gpurun_out/final/fuzz_c.txt
FFPA_FUZZ_SEEDS=35000:37500 timeout 900 python -m pytest tests/test_fwd_gpu.py -m gpu -q -k test_randomized_against_oracle 2>&1 | tail -2 | tee gpurun_out/final/fuzz_e.txt
