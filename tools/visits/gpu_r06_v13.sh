#!/bin/bash
# round 6, visit 13: bench.py measuring roofline.traffic in the run (two rocprofv3 PMC passes behind the timed region) — the driver's command, timed; the bench tests
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
T0=$(date +%s.%N); python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/v13_bench_cfg2.json 2> gpurun_out/r06/v13_bench_cfg2.err; echo "bench exit $? wall $(python -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $T0) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/v13_bench_cfg2.json').read().strip().splitlines()[-1])
r=d['roofline']
print('BENCH cfg2', d['value'], 'TF', d['ms_per_step'], 'ms frac', r['frac'], '| traffic', r['traffic'], '=', round(r['traffic']/r['algorithmic_bytes_per_launch'],3) if r['traffic'] else None, 'x algorithmic | stale', r['traffic_stale'], '| live_failed', r.get('traffic_live_failed'), '|', (r.get('traffic_source') or '')[:60])
PY
timeout 900 python -m pytest tests/test_bench_gpu.py -x -q 2>&1 | tail -3
for w in cfg3 cfg2_causal decode; do python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-sdpa 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d[\"roofline\"]; print(\"BENCH $w\", d[\"value\"], r[\"frac\"], \"traffic\", r[\"traffic\"], r.get(\"traffic_live_failed\"), (r.get(\"traffic_source\") or \"\")[:40])"; done
