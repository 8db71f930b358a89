"""Per-variant means of every PMC counter from a rocprofv3 pass over tools/gpu_ab.py (developer tool).

Usage: python tools/parse_pmc_ab.py <counter_collection.csv> <reps> tag1 tag2 ...
Launch order of gpu_ab.py: one warm-up per tag, then rounds x (tag x reps).
"""
import collections, csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
reps = int(sys.argv[2])
tags = sys.argv[3:]
disp = collections.OrderedDict()
for r in rows:
  if "ffpa_fwd_split" not in r["Kernel_Name"]:
    continue
  d = disp.setdefault(int(r["Dispatch_Id"]), {"ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
  d[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(disp)[len(tags):]
agg = collections.defaultdict(list)
for i in range(0, len(ids) - reps + 1, reps):
  tag = tags[(i // reps) % len(tags)]
  agg[tag] += [disp[k] for k in ids[i:i + reps]]
names = sorted({k for d in disp.values() for k in d})
print("tag".ljust(10) + "".join(n[-18:].rjust(20) for n in names))
for t in tags:
  ds = agg[t]
  print(t.ljust(10) + "".join(f"{sum(d.get(n, 0) for d in ds) / max(len(ds), 1):20.4g}" for n in names))
