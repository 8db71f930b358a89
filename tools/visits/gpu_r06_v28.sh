#!/bin/bash
# round 6, visit 28: more randomized parity on the final library (20179c2c): packed batches with KV ranges FORCED on every launch (prefill launches: per-row-tile ranges),
# dense cases with per-row-tile ranges forced at 2 and 5 ranges
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
sha256sum ffpa_attn_amd/libffpa_attn_hip.so | cut -c1-16 | tee gpurun_out/r06/v28_fuzz.txt
for n in 2 3 7; do
FFPA_VARLEN_FUZZ_SPLITS=$n FFPA_VARLEN_FUZZ=3000:3300 timeout 900 python -m pytest tests/test_varlen_gpu.py -m gpu -q -k test_randomized_packed_batches 2>&1 | grep -E "^E  .*Error|passed|failed" | cut -c1-600 | head -5 | sed "s/^/packed, $n ranges forced, seeds 3000:3300: /" | tee -a gpurun_out/r06/v28_fuzz.txt
done
for n in 2 5; do
FFPA_FUZZ_FLAGS=0x100040 FFPA_FUZZ_SPLITS=$n FFPA_FUZZ_SEEDS=45000:47500 timeout 900 python -m pytest tests/test_fwd_gpu.py -m gpu -q -k test_randomized_against_oracle 2>&1 | grep -E "^E  .*Error|passed|failed" | cut -c1-600 | head -5 | sed "s/^/dense, per-row-tile ranges forced at $n, seeds 45000:47500: /" | tee -a gpurun_out/r06/v28_fuzz.txt
done
