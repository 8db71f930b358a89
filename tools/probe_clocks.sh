#!/bin/bash
# effective shader clock of every stream_probe configuration: GRBM_GUI_ACTIVE (summed over 8 XCDs) / 8 / kernel duration
mkdir -p gpurun_out/probe_clk; rm -rf gpurun_out/probe_clk/*
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $OLDPWD/gpurun_out/probe_clk -o p -- $OLDPWD/tools/probes/bin/stream_probe 1024) > gpurun_out/probe_clk/run.log 2>&1
python3 - <<'PY'
import csv, collections, glob
rows = list(csv.DictReader(open(glob.glob('gpurun_out/probe_clk/**/*counter_collection.csv', recursive=True)[0])))
d = collections.OrderedDict()
for r in rows:
    e = d.setdefault(r['Dispatch_Id'], {'k': r['Kernel_Name'], 'ms': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6})
    e[r['Counter_Name']] = float(r['Counter_Value'])
for i, e in d.items():
    ghz = e.get('GRBM_GUI_ACTIVE', 0) / 8 / (e['ms'] * 1e6)
    w = e.get('SQ_WAVE_CYCLES', 1)
    print('CLK %-44s %8.3f ms  clock %.2f GHz  mfma_busy %.2f  wait_inst %.2f  active %.2f  wait_any %.2f' % (e['k'][8:52], e['ms'], ghz, e.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * w), e.get('SQ_WAIT_INST_ANY', 0) / w, e.get('SQ_ACTIVE_INST_ANY', 0) / w, e.get('SQ_WAIT_ANY', 0) / w))
PY
