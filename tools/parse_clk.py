"""Per-variant effective clock from a rocprofv3 pmc run of tools/gpu_ab.py (developer tool)."""
import csv, sys, collections
tags = sys.argv[2:]
rows = list(csv.DictReader(open(sys.argv[1])))
disp = collections.OrderedDict()
for r in rows:
  if "ffpa_fwd_" not in r["Kernel_Name"] or "merge" in r["Kernel_Name"]:
    continue
  d = disp.setdefault(int(r["Dispatch_Id"]), {"dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
  d[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(disp)
# order of launches in gpu_ab: one warm-up per tag, then rounds x (tag x reps)
n = len(tags)
warm = ids[:n]
rest = ids[n:]
reps = 3
agg = collections.defaultdict(list)
i = 0
while i + reps <= len(rest):
  tag = tags[(i // reps) % n]
  for j in range(reps):
    agg[tag].append(disp[rest[i + j]])
  i += reps
for t in tags:
  ds = agg[t]
  dur = sum(d["dur"] for d in ds) / len(ds)
  cyc = sum(d["GRBM_GUI_ACTIVE"] for d in ds) / len(ds) / 8
  wave = sum(d["SQ_WAVE_CYCLES"] for d in ds) / len(ds)
  print(f"{t:10s} dur {dur:6.3f} ms  cycles/XCD {cyc/1e6:6.3f} M  clock {cyc/dur/1e6:5.3f} GHz  mfma_busy {sum(d['SQ_VALU_MFMA_BUSY_CYCLES'] for d in ds)/len(ds)/(wave*4):.3f}"
        f"  wait_any {sum(d['SQ_WAIT_ANY'] for d in ds)/len(ds)/wave:.3f} wait_inst {sum(d['SQ_WAIT_INST_ANY'] for d in ds)/len(ds)/wave:.3f} active {sum(d['SQ_ACTIVE_INST_ANY'] for d in ds)/len(ds)/wave:.3f}")
