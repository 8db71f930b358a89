#!/bin/bash
# round 6, visit 3: DecodeStep (the graph-replayed decode step) — parity tests, then the decode bench line through it next to the per-call form
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_host_path_gpu.py -x -q > gpurun_out/r06/v3_pytest.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/r06/v3_pytest.log
timeout 600 python bench.py --workload decode --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06/v3_decode.json 2> gpurun_out/r06/v3_decode.err; echo "bench exit $?"; tail -3 gpurun_out/r06/v3_decode.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/v3_decode.json').read().strip().splitlines()[-1])
print('DECODE ms_per_step', d['ms_per_step'], 'kernel_ms_avg', d['roofline']['kernel_ms_avg'], 'frac', d['roofline']['frac'], 'eager', d.get('eager_api'), 'steady', d.get('steady_state',{}).get('ms_per_step'), 'graph', {k:v.get('ms_per_step') for k,v in d.get('graph_replay',{}).items() if isinstance(v,dict)})
PY
