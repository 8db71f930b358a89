#!/bin/bash
# round 6, visit 18: the packed-decode bench line (bench.py --workload varlen_decode) + its test; the whole GPU suite on the library with ABI 6
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
python bench.py --workload varlen_decode --steps 20 --warmup 5 > gpurun_out/r06/v18_bench_varlen_decode.json 2> gpurun_out/r06/v18_bench_varlen_decode.err; echo "bench exit $?"; tail -3 gpurun_out/r06/v18_bench_varlen_decode.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/v18_bench_varlen_decode.json').read().strip().splitlines()[-1])
r=d['roofline']
print('BENCH varlen_decode', d['value'], 'TF', d['ms_per_step'], 'ms |', r['achieved'], 'GB/s frac', r['frac'], '| steady', d['steady_state'], '| traffic', r['traffic'], r.get('traffic_live_failed'))
print(' plan', d['plan'], r['kernel'])
print(' legs', json.dumps(d.get('other_launches')))
print(' loop', json.dumps(d.get('per_sequence_loop')), 'sdpa', d.get('sdpa_gpu_ms'), d.get('max_abs_err_vs_sdpa'))
PY
timeout 3000 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -6
