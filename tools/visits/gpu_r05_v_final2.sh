#!/bin/bash
# After the bench restructure (gather leg last, under a watchdog): the default bench line, the bench / sharding GPU tests, the causal-gap decomposition, and the
# randomized parity sweep on the final library with fresh seeds.
export TMPDIR=/tmp
O=gpurun_out/final2; mkdir -p $O
python -c "from ffpa_attn_amd import hip; print('LIB', hip.build_identity() if hasattr(hip,'build_identity') else '')" 2>/dev/null | tail -1
sha256sum ffpa_attn_amd/libffpa_attn_hip.so | cut -c1-16
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/final2/bench_default.json').read().strip().splitlines()[-1])
print('BENCH', d['value'], d['unit'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'stale', d['roofline']['traffic_stale'], 'cpu', d.get('cpu_baseline',{}).get('value'), 'lib', d['build'].get('lib_sha16'))
PY
timeout 900 python -m pytest tests/test_bench_gpu.py tests/test_sharding_gpu.py tests/test_host_path_gpu.py -m gpu -q 2>&1 | tail -3
timeout 300 python tools/gpu_causal_gap.py 2>&1 | grep CAUSALGAP | tee $O/causal_gap.txt
FFPA_FUZZ_SEEDS=3000:5500 timeout 600 python -m pytest tests/test_fwd_gpu.py -m gpu -q -k test_randomized_against_oracle 2>&1 | tail -2 | tee $O/fuzz_a.txt
FFPA_FUZZ_FLAGS=0x1000 FFPA_FUZZ_SEEDS=5500:8000 timeout 600 python -m pytest tests/test_fwd_gpu.py -m gpu -q -k test_randomized_against_oracle 2>&1 | tail -2 | tee $O/fuzz_b.txt
