#!/bin/bash
# round 6, visit 4: the library with the plan code as a rule table (plans identical on a 26k-call corpus) + FFPA_FLAG_DETERMINISTIC: the whole GPU suite;
# host time of a decode step through DecodeStep vs the plain call; the decode bench line both ways
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r06/v4_pytest.log 2>&1; echo "pytest exit $?"; tail -6 gpurun_out/r06/v4_pytest.log
timeout 600 python tools/gpu_host_overhead.py > gpurun_out/r06/v4_host.txt 2>&1; echo "host exit $?"; grep HOST gpurun_out/r06/v4_host.txt
for mode in 0 1; do
  FFPA_BENCH_DECODE_EAGER=$mode timeout 600 python bench.py --workload decode --steps 20 --warmup 5 --no-cpu-baseline --no-sdpa > gpurun_out/r06/v4_decode_eager$mode.json 2> gpurun_out/r06/v4_decode.err; echo "bench exit $?"
  python - <<PY
import json
d=json.loads(open('gpurun_out/r06/v4_decode_eager$mode.json').read().strip().splitlines()[-1])
print('DECODE eager=$mode step:', d['config']['step'][:40], '| ms_per_step', d['ms_per_step'], 'kernel_ms_avg', d['roofline']['kernel_ms_avg'], 'frac', d['roofline']['frac'], 'eager_api', (d.get('eager_api') or {}).get('ms_per_step'), 'steady', d.get('steady_state',{}).get('ms_per_step'), 'graph', {k:v.get('ms_per_step') for k,v in d.get('graph_replay',{}).items() if isinstance(v,dict)})
PY
done
