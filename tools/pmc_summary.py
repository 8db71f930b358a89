"""Summarise rocprofv3 --pmc passes of bench.py into the JSON committed under profiles/ (developer tool).

Usage: python tools/pmc_summary.py <dir with *counter_collection.csv> <out.json> [note]

Per counter: mean over the ffpa_fwd_split_d_kernel dispatches of every pass found.  Derived values follow
/opt/skills/guides/MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles,
SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per 32x32x16 MFMA), GRBM_GUI_ACTIVE is summed over the 8 XCDs,
FETCH_SIZE is KiB and reads half of a wide coalesced stream on gfx950 (corrected x2 here).
"""
import collections, csv, glob, json, os, sys


def main():
  src, out = sys.argv[1], sys.argv[2]
  note = sys.argv[3] if len(sys.argv) > 3 else ""
  vals = collections.defaultdict(list)
  durs, grbm_durs = [], []
  for path in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
    seen = set()
    for r in csv.DictReader(open(path)):
      if "ffpa_fwd_split_d_kernel" not in r["Kernel_Name"]:
        continue
      vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
      dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
      if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        grbm_durs.append(dur)
      if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        durs.append(dur)
  res = {
      "source": "rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sdpa "
                "(one pass per counter group, tools/gpu_round.sh stages pmcsq + pmcfetch), MI355X. " + note,
      "units": "SQ_WAVE_CYCLES/SQ_WAIT_*/SQ_ACTIVE_INST_* are quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES are cycles (=32 x N_mfma); "
               "GRBM_GUI_ACTIVE is summed over 8 XCDs; FETCH_SIZE in KiB and reads 1/2 of a wide coalesced stream on gfx950 "
               "(MI355X_MICROARCH.md, HBM section)",
  }
  for k, v in sorted(vals.items()):
    res[k] = {"per_dispatch_mean": sum(v) / len(v), "dispatches": len(v)}
  m = lambda k: res[k]["per_dispatch_mean"] if k in res else None
  d = {"kernel_ms_mean_under_pmc": sum(durs) / len(durs) if durs else None}
  if m("SQ_WAVE_CYCLES"):
    w = m("SQ_WAVE_CYCLES")
    if m("SQ_VALU_MFMA_BUSY_CYCLES"):
      d["mfma_busy_fraction_of_simd_cycles"] = m("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * w)
    for k, name in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_any_frac"), ("SQ_ACTIVE_INST_ANY", "active_inst_any_frac")):
      if m(k):
        d[name] = m(k) / w
  if m("SQ_LDS_BANK_CONFLICT") is not None:
    d["lds_bank_conflict_cycles"] = m("SQ_LDS_BANK_CONFLICT")
  if m("GRBM_GUI_ACTIVE"):
    d["cycles_per_xcd"] = m("GRBM_GUI_ACTIVE") / 8
    if grbm_durs:  # clock of the pass that carried GRBM_GUI_ACTIVE (profiled passes clock lower than plain runs)
      d["effective_clock_ghz_in_that_pass"] = d["cycles_per_xcd"] / (sum(grbm_durs) / len(grbm_durs)) / 1e6
  if m("FETCH_SIZE"):
    d["fetch_size_KiB"] = m("FETCH_SIZE")
    d["hbm_read_bytes_corrected_x2"] = m("FETCH_SIZE") * 1024 * 2
    if m("WRITE_SIZE"):
      d["hbm_write_bytes"] = m("WRITE_SIZE") * 1024  # KiB; uncalibrated on gfx950 per the guide
    d["algorithmic_bytes_Q+K+V+O+LSE"] = 4 * 32 * 8192 * 512 * 2 + 32 * 8192 * 4
  res["derived"] = d
  json.dump(res, open(out, "w"), indent=1)
  print(json.dumps(d, indent=1))


if __name__ == "__main__":
  main()
