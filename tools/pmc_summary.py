"""Summarise rocprofv3 --pmc passes of bench.py into the JSON committed under profiles/ (developer tool).

Usage: python tools/pmc_summary.py <dir with *counter_collection.csv> <out.json> [note] [--kernel SUBSTR]

Per counter: mean over the dispatches of the dominant kernel (default: the prefill kernels, either MFMA-shape build) of every pass found.
Derived values follow /opt/skills/guides/MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per 32x32x16 MFMA, 16 per 16x16x32), GRBM_GUI_ACTIVE is summed over the 8 XCDs,
FETCH_SIZE is KiB and reads half of a wide coalesced stream on gfx950 (corrected x2 here).  The workload's algorithmic
bytes / FLOPs and the kernel's L2->LDS operand stream (profiles/NOTES.md section 3) are taken from the bench line in
<dir>/*.log when one is found, so that request counters can be read as bytes per request.
"""
import collections, csv, glob, json, os, re, sys


def bench_line(src):
  # (the --kernel-trace pass's own bench line first: its steps / warm-ups are what the trace statistics below count)
  for path in sorted(glob.glob(os.path.join(src, "*.log")), key=lambda p: (os.path.basename(p) != "trace.log", p)):
    for line in open(path, errors="replace"):
      line = line.strip()
      if line.startswith("{") and '"roofline"' in line:
        try:
          return json.loads(line)
        except ValueError:
          pass
  return None


def trace_stats(src, kernel, warmup, steps=None):
  """Durations (ms) of the dominant kernel's dispatches in the --kernel-trace pass of the same command (<dir>/**/trace_kernel_trace.csv),
  in dispatch order, warm-up launches dropped: median / mean / min / max / count — what profiles/ must reproduce of the bench line."""
  for path in sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)):
    if not os.path.basename(path).startswith("trace"):  # (the --stats pass is written with -o trace; the PMC passes carry their own names)
      continue
    rows = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in csv.DictReader(open(path))
            if re.search(kernel, r["Kernel_Name"])]
    if not rows:
      continue
    d = [x for _, x in sorted(rows)][warmup:(warmup + steps) if steps else None]  # the timed region's launches only
    if not d:
      continue
    sd = sorted(d)
    return {"file": os.path.relpath(path, src), "dispatches_counted": len(d), "warmup_dispatches_dropped": warmup, "median_ms": sd[len(sd) // 2],
            "mean_ms": sum(d) / len(d), "min_ms": sd[0], "max_ms": sd[-1]}
  return None


def main():
  argv = sys.argv[1:]
  kernel = r"ffpa_fwd_(split_d|m16w?|m16_pair|m16_varlen)_kernel"  # (regex) the prefill kernels: 32x32x16-MFMA build / 16x16x32-MFMA build (+ its wide-row tile, its paired-tile and packed-sequence kernels)
  if "--kernel" in argv:
    i = argv.index("--kernel")
    kernel = argv[i + 1]
    del argv[i:i + 2]
  src, out = argv[0], argv[1]
  note = argv[2] if len(argv) > 2 else ""
  vals = collections.defaultdict(list)
  durs, grbm_durs = [], []
  names = set()
  for path in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
    seen = set()
    for r in csv.DictReader(open(path)):
      if not re.search(kernel, r["Kernel_Name"]):
        continue
      names.add(re.sub(r"<.*", "", r["Kernel_Name"]).replace("void ", "").replace("ffpa::", ""))
      vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
      dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
      if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        grbm_durs.append(dur)
      if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        durs.append(dur)
  res = {
      "source": "rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --workload W --steps 3 --warmup 1 --no-cpu-baseline --no-sdpa "
                "(one pass per counter group, tools/gpu_round.sh stage wprof), MI355X. " + note,
      "units": "SQ_WAVE_CYCLES/SQ_WAIT_*/SQ_ACTIVE_INST_* are quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES are cycles (=32 x N_mfma); "
               "GRBM_GUI_ACTIVE is summed over 8 XCDs; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950 "
               "(MI355X_MICROARCH.md, HBM section); *_sum counters are summed over all instances of the block",
      "kernel": ", ".join(sorted(names)) or kernel,
  }
  for k, v in sorted(vals.items()):
    res[k] = {"per_dispatch_mean": sum(v) / len(v), "dispatches": len(v)}
  m = lambda k: res[k]["per_dispatch_mean"] if k in res else None
  d = {"kernel_ms_mean_under_pmc": sum(durs) / len(durs) if durs else None}
  if m("SQ_WAVE_CYCLES"):
    w = m("SQ_WAVE_CYCLES")
    if m("SQ_VALU_MFMA_BUSY_CYCLES"):
      d["mfma_busy_fraction_of_simd_cycles"] = m("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * w)
    for k, name in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_any_frac"), ("SQ_ACTIVE_INST_ANY", "active_inst_any_frac"),
                    ("SQ_WAIT_INST_LDS", "wait_inst_lds_frac"), ("SQ_ACTIVE_INST_LDS", "active_inst_lds_frac"),
                    ("SQ_ACTIVE_INST_VMEM", "active_inst_vmem_frac"), ("SQ_ACTIVE_INST_VALU", "active_inst_valu_frac")):
      if m(k) is not None:
        d[name] = m(k) / w
    if m("SQ_INST_LEVEL_VMEM") is not None:
      d["vmem_instructions_in_flight_per_wave_mean"] = m("SQ_INST_LEVEL_VMEM") / w
    if m("SQ_INST_LEVEL_LDS") is not None:
      d["lds_instructions_in_flight_per_wave_mean"] = m("SQ_INST_LEVEL_LDS") / w
  if m("SQ_LDS_BANK_CONFLICT") is not None:
    d["lds_bank_conflict_cycles"] = m("SQ_LDS_BANK_CONFLICT")
  if m("SQ_LDS_IDX_ACTIVE") and m("SQ_BUSY_CYCLES"):
    d["lds_idx_active_over_sq_busy_cycles"] = m("SQ_LDS_IDX_ACTIVE") / m("SQ_BUSY_CYCLES")
  if m("GRBM_GUI_ACTIVE"):
    d["cycles_per_xcd"] = m("GRBM_GUI_ACTIVE") / 8
    if grbm_durs:  # clock of the pass that carried GRBM_GUI_ACTIVE (profiled passes clock lower than plain runs)
      d["effective_clock_ghz_in_that_pass"] = d["cycles_per_xcd"] / (sum(grbm_durs) / len(grbm_durs)) / 1e6
  if m("FETCH_SIZE"):
    d["fetch_size_KiB"] = m("FETCH_SIZE")
    d["hbm_read_bytes_corrected_x2"] = m("FETCH_SIZE") * 1024 * 2
  if m("WRITE_SIZE"):
    d["hbm_write_bytes"] = m("WRITE_SIZE") * 1024  # KiB; uncalibrated on gfx950 per the guide
  if m("TCC_HIT_sum") is not None and m("TCC_MISS_sum") is not None and m("TCC_HIT_sum") + m("TCC_MISS_sum") > 0:
    d["l2_hit_rate"] = m("TCC_HIT_sum") / (m("TCC_HIT_sum") + m("TCC_MISS_sum"))
  line = bench_line(src)
  # provenance: which binary, which tree, which clock (a profile that does not name its binary cannot back a bench line)
  prov = {"git_head": os.environ.get("FFPA_GIT_HEAD") or (line or {}).get("build", {}).get("git_head"),
          "lib_sha16": (line or {}).get("build", {}).get("lib_sha16"), "lib_version": (line or {}).get("build", {}).get("lib_version"),
          "bench_kernel": (line or {}).get("roofline", {}).get("kernel"), "effective_clock_ghz_under_pmc": d.get("effective_clock_ghz_in_that_pass")}
  res["provenance"] = prov
  ts = trace_stats(src, kernel, int((line or {}).get("warmup", os.environ.get("PMC_TRACE_WARMUP", "5"))), (line or {}).get("steps"))
  if ts:
    d["kernel_trace"] = ts
    roof = (line or {}).get("roofline", {})
    work = roof.get("flops_per_launch") or roof.get("bytes_per_launch")
    if work and roof.get("peak"):
      unit = 1e12 if roof.get("bound") == "mfma" else 1e9
      d["kernel_trace"]["frac_of_peak_at_median"] = work / (ts["median_ms"] * 1e-3) / unit / roof["peak"]
      d["kernel_trace"]["frac_of_peak_at_mean"] = work / (ts["mean_ms"] * 1e-3) / unit / roof["peak"]
      d["kernel_trace"]["bench_line_frac"] = roof.get("frac")
      d["kernel_trace"]["bench_line_kernel_ms_avg"] = roof.get("kernel_ms_avg")
  if line:
    wl = line.get("config", {}).get("workload", "")
    d["workload"] = wl
    g = re.search(r"B=(\d+) Hq=(\d+) Hkv=(\d+) Nq=(\d+) Nkv=(\d+) D=(\d+)", wl)
    if g:
      B, Hq, Hkv, Nq, Nkv, D = (int(x) for x in g.groups())
      d["algorithmic_bytes_Q+K+V+O+LSE"] = 2 * D * (2 * B * Hq * Nq + 2 * B * Hkv * Nkv) + 4 * B * Hq * Nq
      br = 128 if D <= 512 else 64
      stream = B * Hq * ((Nq + br - 1) // br) * 2 * Nkv * D * 2 + B * Hq * Nq * D * 2
      d["l2_to_lds_operand_stream_bytes"] = stream  # every row tile streams K and V once (+ Q once)
      if d.get("cycles_per_xcd"):
        d["operand_stream_bytes_per_clk_per_cu"] = stream / 256 / d["cycles_per_xcd"]
      for k in ("TCP_TCC_READ_REQ_sum", "TCC_REQ_sum", "TCC_READ_sum", "TA_BUFFER_READ_LDS_WAVEFRONTS_sum"):
        if m(k):
          d["operand_stream_bytes_per_" + k] = stream / m(k)
      if d["kernel_ms_mean_under_pmc"]:
        d["operand_stream_TBps_under_pmc"] = stream / (d["kernel_ms_mean_under_pmc"] * 1e-3) / 1e12
    elif (line.get("roofline") or {}).get("algorithmic_bytes_per_launch"):  # (the packed-sequence workload: no single B / Nq / Nkv — the line carries the figure)
      d["algorithmic_bytes_Q+K+V+O+LSE"] = line["roofline"]["algorithmic_bytes_per_launch"]
  res["derived"] = d
  json.dump(res, open(out, "w"), indent=1)
  print(json.dumps(d, indent=1))


if __name__ == "__main__":
  main()
