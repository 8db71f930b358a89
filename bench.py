"""bench.py — the reference's headline benchmark (and its whole forward case matrix) on MI355X.

    python bench.py                                  # BASELINE metric: cfg2 = B1 H32 N8192 D512 bf16, 1 GPU
    python bench.py --workload cfg3 | cfg4_mask | cfg4_offset0 | cross | gqa | attn_mask | dropout | non_aligned | decode | ...
    python bench.py --sweep [--sweep-dims 320,512,1024]   # the reference bench's one-shot case table (one process, one table per head dim)
    python bench.py --gpus N [--workload cfg5] [--gather]  # N > 1 without a launcher: spawns its own N ranks (free port, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N [--workload cfg5] [--gather]     # ... or under torchrun (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env)

One "step" = one pass of the hot path (ffpa_attn_func -> ffpa_attn::_fwd_hip -> C-ABI -> HIP kernel) over one
synthetic batch already resident in HBM.  FLOPs = 4*B*Hq*D*valid_pairs, the reference's own model
(src/ffpa_attn/cli/_flops.py:37-53); inputs are seed-0 randn, q then k then v (cli/_runner_fwd.py:344-347).
The workloads are BASELINE.json's configs plus the reference bench's cases at its defaults (H=32, N=8192, D=512:
self / cross 1024xN / decode / gqa H/4 / causal / attn-mask [1,1,1,Nkv]*0.25 / dropout 0.1 / non-aligned N-1 with
H/4 heads; cli/_runner_fwd.py:599-672).

N > 1: the path is embarrassingly parallel over (batch, kv-head) units (ffpa_attn_amd/sharding.py) and every rank's
block is BORN sharded (per-unit seeds, nothing replicated), no data-path collective:
  * default workloads: every rank owns one workload-shaped batch element -> "scaling": "weak" (8 x cfg2 is config 5);
  * --workload cfg5: the fixed global problem B=8 H=32 N=8192 D=512 (256 units) is split over the ranks ->
    "scaling": "strong"; per-rank TFLOPS are listed next to the aggregate;
  * --gather puts the gather of O a caller wanting the full tensor on every rank would pay INTO the timed step, in --gather-chunks pieces
    sent point-to-point over RCCL while the next piece computes (sharding.attend_and_gather_units); without it the same K steps are timed a
    second time with the gather, after the contract figure and under a watchdog (guarded_extra_leg), and reported as "with_gather".
Timing: barrier + synchronize on both sides of exactly K steps, max over ranks.

Rank 0 prints ONE JSON line.  `roofline` is the dominant kernel against the dense bf16 MFMA peak (HBM peak for the
decode workload), its average launch duration measured by HIP events on the launch stream; `roofline.traffic` is the
HBM bytes per launch from the committed rocprofv3 PMC pass of the same command (`traffic_source` names the file; null
when no such pass is committed) — a profile artefact, not a measurement of this very run.  `cpu_baseline` is the
reference's CPU path for this op (PyTorch CPU SDPA — the reference has no CPU kernel, ffpa_attn_interface.py:165-176)
timed on this box's host cores on a bounded sample.  `ref_protocol` repeats the reference bench's own timing protocol
(2 warm-ups, 10 iterations, perf_counter + synchronize: cli/_runner_fwd.py:84-103).
"""

from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

from ffpa_attn_amd import ffpa_attn_func  # noqa: E402
from ffpa_attn_amd.flops import attention_fwd_flops  # noqa: E402

# Dense bf16 MFMA peak of one MI355X, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters":
# 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz ~= 2.5 PFLOP/s (measured 2470-2495 TF with 32x32x16 on zero operands).
MFMA_BF16_PEAK_TFLOPS = 2500.0
HBM_PEAK_GBPS = 8000.0  # same guide: 8 TB/s spec (about 6.3 TB/s achievable)

BASELINE_METRIC = "attention fwd TFLOPS + max-abs-err vs SDPA, bf16 B=1 H=32 N=8192 D=512"


def _w(B, Hq, Hkv, Nq, Nkv, D, **kw):
  d = dict(B=B, Hq=Hq, Hkv=Hkv, Nq=Nq, Nkv=Nkv, D=D, causal=False, mask=None, dropout=0.0, via="api", bound="mfma", note="")
  d.update(kw)
  return d


WORKLOADS = {
  # BASELINE.json configs
  "cfg2": _w(1, 32, 32, 8192, 8192, 512, note="BASELINE configs[1], the headline shape"),
  "cfg3": _w(1, 32, 32, 8192, 8192, 1024, note="BASELINE configs[2], max head dim"),
  "cfg4_mask": _w(2, 32, 8, 8192, 2048, 320, mask="tril_bool", note="BASELINE configs[3] as specified: GQA cross-attention + causal mask, "
                  "ffpa_attn_func(attn_mask=tril(ones(Nq,Nkv,bool)), enable_gqa=True); FLOPs count the visible pairs"),
  "cfg4_offset0": _w(2, 32, 8, 8192, 2048, 320, causal=True, via="op_offset0", note="configs[3] through the op's structured top-left causal mask "
                     "(causal_offset=0 = SDPA's is_causal for Nq > Nkv)"),
  "cfg4_nomask": _w(2, 32, 8, 8192, 2048, 320, note="configs[3] shape without the mask (dense FLOPs)"),
  "cfg5": _w(8, 32, 32, 8192, 8192, 512, note="BASELINE configs[4]: the global B=8 problem split over the ranks (strong scaling)"),
  # the reference bench's case matrix at its defaults (cli/_runner_fwd.py:599-672)
  "cfg2_causal": _w(1, 32, 32, 8192, 8192, 512, causal=True, note="reference bench case 'causal'"),
  "cross": _w(1, 32, 32, 1024, 8192, 512, note="reference bench case 'cross-attn' (Nq = 1024)"),
  "gqa": _w(1, 32, 8, 8192, 8192, 512, note="reference bench case 'gqa' (Hkv = H/4)"),
  "gqa_causal": _w(1, 32, 8, 8192, 8192, 512, causal=True, note="the reference bench's 'gqa' and 'causal' cases together (LLM-style prefill): the launch the head-chunk workgroup order is for"),
  "prompt_tp8": _w(1, 8, 8, 4096, 4096, 512, causal=True, one_range_leg=True,
                   note="a whole 4096-token prompt on a tensor-parallel shard of 8 heads (causal): 256 row tiles = ONE round of workgroups — the launch whose row tiles split their own visible "
                        "KV tiles (ffpa_capi.hip pick_tile_ranges); `other_launches`: the same call with one KV range per row tile (paired row tiles: what ran before)"),
  "attn_mask": _w(1, 32, 32, 8192, 8192, 512, mask="key_bias", note="reference bench case 'attn-mask': additive [1,1,1,Nkv] randn*0.25 (cli/_runner_fwd.py:75-81)"),
  "dropout": _w(1, 32, 32, 8192, 8192, 512, dropout=0.1, note="reference bench case 'dropout' (p = 0.1, in-kernel Philox)"),
  "non_aligned": _w(1, 8, 8, 8191, 8191, 512, note="reference bench case 'non-aligned' (N-1, H/4 heads)"),
  "decode": _w(1, 32, 32, 1, 8192, 512, bound="hbm", note="reference bench case 'decode-attn' (Nq = 1: split-KV kernel + LSE merge; HBM-bound)"),
  # packed sequences (ffpa_attn_varlen_func: the reference's CuTe-DSL-only entry point, one launch here): 8 causal self-attention sequences, 16384 tokens
  "varlen": _w(8, 32, 8, 4864, 4864, 512, causal=True, via="varlen", lens=(4096, 512, 2048, 1024, 3072, 256, 4864, 512),
               note="packed THD batch through ffpa_attn_varlen_func: 8 causal sequences of 256 ... 4864 tokens (16384 in all), GQA 32 / 8, one launch"),
  # ... and a packed DECODE batch (continuous batching: one new token per sequence against ragged KV lengths): HBM-bound, K + V streamed once
  "varlen_decode": _w(32, 32, 8, 1, 16384, 512, causal=True, via="varlen_decode", bound="hbm", kv_range=(1024, 16384),
                      note="packed decode batch through the packed-sequence call: 32 sequences x 1 token against ragged KV lengths 1024 ... 16384 (seeded), GQA 32 / 8: heads of a KV "
                           "group packed into tile rows, KV ranges split + merged, non-temporal K / V fetch; HBM-bound"),
}


def build_identity() -> dict:
  """What binary produced the numbers: sha256 of the loaded library (first 16 hex digits), its version string, and the git head of the
  tree it was built from when that is knowable (`git rev-parse`; the GPU box receives a snapshot without .git, where
  `tools/gpu_round.sh` exports FFPA_GIT_HEAD)."""
  from ffpa_attn_amd import hip

  out = {"lib": os.path.relpath(hip.LIB_PATH, ROOT)}
  try:
    out["lib_sha16"] = hashlib.sha256(open(hip.LIB_PATH, "rb").read()).hexdigest()[:16]
    out["lib_version"] = hip.load_library().ffpa_attn_version().decode()
  except OSError:
    pass
  head = os.environ.get("FFPA_GIT_HEAD")
  if not head:
    try:
      head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip()
      dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True, timeout=5).stdout.strip()
      if head and dirty:
        head += "+dirty"
    except (OSError, subprocess.SubprocessError):
      head = ""
  if head:
    out["git_head"] = head
  return out


def planned_kernel(w: dict, q, k, v, mask, scale: float) -> dict:
  """The launch plan of this workload's call, from the C-ABI (ffpa_attn_fwd_plan / ffpa_attn_fwd_kernel): the kernel the roofline
  object is about is whatever the library says it launches — no copy of the dispatch rule lives here."""
  from ffpa_attn_amd import hip

  plan = {}
  bias = None if (w["via"] == "op_offset0" or mask is None) else (mask if mask.dim() == 4 else mask.view(1, 1, *mask.shape))
  hip.forward(q, k, v, bias, bool(w["causal"]), scale, causal_offset=0 if w["via"] == "op_offset0" else None, dropout_p=w["dropout"],
              philox_seed=1, return_lse=False, plan_out=plan)
  return plan


def metric_name(name: str, w: dict) -> str:
  if name == "cfg2":
    return BASELINE_METRIC
  heads = f"H={w['Hq']}" if w["Hq"] == w["Hkv"] else f"Hq={w['Hq']}/Hkv={w['Hkv']}"
  seq = f"N={w['Nq']}" if w["Nq"] == w["Nkv"] else f"Nq={w['Nq']} Nkv={w['Nkv']}"
  tags = [t for t in ("causal" if w["causal"] else "", {"tril_bool": "bool causal mask", "key_bias": "additive key bias"}.get(w["mask"], ""),
                      f"dropout {w['dropout']}" if w["dropout"] else "") if t]
  return f"attention fwd TFLOPS + max-abs-err vs SDPA, bf16 B={w['B']} {heads} {seq} D={w['D']}" + (" " + " + ".join(tags) if tags else "") + f" [{name}]"


def valid_pairs_flops(w: dict, B: int) -> int:
  causal = w["causal"] or w["mask"] == "tril_bool"
  if w["via"] == "op_offset0" or w["mask"] == "tril_bool":  # top-left alignment: key <= row
    pairs = sum(min(w["Nkv"], r + 1) for r in range(w["Nq"]))
    return 4 * B * w["Hq"] * w["D"] * pairs
  return attention_fwd_flops(B, w["Hq"], w["Nq"], w["Nkv"], w["D"], causal)


def algorithmic_bytes(w: dict, B: int) -> int:
  return 2 * w["D"] * (2 * B * w["Hq"] * w["Nq"] + 2 * B * w["Hkv"] * w["Nkv"]) + 4 * B * w["Hq"] * w["Nq"]


def make_mask(w: dict, dtype, device, seed: int = 0):
  if w["mask"] == "tril_bool":
    return torch.ones(w["Nq"], w["Nkv"], dtype=torch.bool, device=device).tril()
  if w["mask"] == "key_bias":
    torch.manual_seed(seed + 1)
    return torch.randn(1, 1, 1, w["Nkv"], dtype=dtype, device=device) * 0.25
  return None


def measured_traffic(workload: str, lib_sha16: str | None):
  """(bytes, source, stale): HBM bytes per launch from the committed rocprofv3 PMC pass of this workload's bench command
  (`rocprofv3 --pmc FETCH_SIZE` x2 — gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md §HBM — plus
  `--pmc WRITE_SIZE`; tools/gpu_round.sh stage wprof, summarised by tools/pmc_summary.py).  bench.py cannot profile
  itself, so this is a profile artefact named by `traffic_source` — and it is only quoted when the profile's
  `provenance.lib_sha16` IS the library this run loaded: a profile of another binary gives (None, its path, True) and the line
  says `traffic_stale`.  (None, None, False) when no profile is committed."""
  import glob
  import re

  found = [p_ for p_ in glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_bench_{workload}_pmc.json")) if re.fullmatch(rf"r\d\d_bench_{re.escape(workload)}_pmc\.json", os.path.basename(p_))]
  for path in sorted(found, reverse=True):  # the newest round's profile first
    if os.path.exists(path):
      try:
        doc = json.load(open(path))
        d = doc["derived"]
        if lib_sha16 is None or doc.get("provenance", {}).get("lib_sha16") != lib_sha16:
          return None, os.path.relpath(path, ROOT), True
        return int(d["hbm_read_bytes_corrected_x2"] + d.get("hbm_write_bytes", 0)), os.path.relpath(path, ROOT), False
      except (KeyError, ValueError):
        pass
  return None, None, False


def live_traffic(workload: str, timeout_s: float = 150.0):
  """(bytes, source) or (None, reason): HBM bytes per launch of the dominant kernel MEASURED IN THIS RUN — two short `rocprofv3 --kernel-trace --pmc` passes
  (FETCH_SIZE, then WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md section 'rocprofv3 PMC slots') of this very script with the same workload on the
  same GPU and library, 1 warm-up + 3 steps each, counters averaged over the kernel's dispatches; FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of
  a wide coalesced stream at 64 bytes: the guide's HBM section; WRITE_SIZE as read).  Runs after the timed region, never inside it.  Any failure — no rocprofv3,
  a pass that times out, counters the profiler does not know — returns (None, why) and the line quotes the committed profile instead."""
  import csv
  import glob
  import re
  import shutil
  import signal
  import subprocess
  import tempfile

  prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
  if prof is None:
    return None, "rocprofv3 not found"
  kernel = re.compile(r"ffpa_fwd_(split_d|m16w?|m16_pair|m16_varlen)_kernel")
  out = tempfile.mkdtemp(prefix="ffpa_bench_pmc_", dir="/tmp")
  got = {}
  try:
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
      cmd = [prof, "--kernel-trace", "--output-format", "csv", "--pmc", counter, "-d", out, "-o", counter.lower(), "--", sys.executable, os.path.abspath(__file__),
             "--workload", workload, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-sdpa", "--no-steady", "--no-ref-protocol", "--no-live-traffic"]
      pr = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
      try:
        rc = pr.wait(timeout=timeout_s / 2)
      except subprocess.TimeoutExpired:
        os.killpg(pr.pid, signal.SIGKILL)  # (the process group this call started: the profiler and the bench under it)
        pr.wait()
        return None, f"rocprofv3 --pmc {counter} did not finish within {timeout_s / 2:g} s"
      if rc != 0:
        return None, f"rocprofv3 --pmc {counter} exited with {rc}"
      vals = []
      for path in glob.glob(os.path.join(out, "**", f"{counter.lower()}*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
          if r.get("Counter_Name") == counter and kernel.search(r.get("Kernel_Name", "")):
            vals.append(float(r["Counter_Value"]))
      if not vals:
        return None, f"no {counter} rows for the kernel in the profiler's output"
      got[counter] = sum(vals) / len(vals)
  except Exception as e:  # noqa: BLE001 — informative only
    return None, f"{type(e).__name__}: {e}"[:200]
  finally:
    shutil.rmtree(out, ignore_errors=True)
  total = int(got["FETCH_SIZE"] * 1024 * 2 + got["WRITE_SIZE"] * 1024)
  return total, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --workload " + workload +
                 " --steps 3 --warmup 1` on this GPU and library (FETCH_SIZE KiB x 2: gfx950 counts a wide stream's 128-byte requests as 64; WRITE_SIZE KiB as read; mean over the kernel's dispatches)")


class DeviceTelemetry:
  """Shader clock and socket power of the benchmarked GPU, sampled from its sysfs hwmon node by a thread while the timed region runs,
  plus what the device says about itself (CU count, maximum shader clock): the numbers that explain why two boxes of a pool give one
  binary +- 5 % (SURVEY.md section 8d asks for them next to every figure).  Everything is best effort: a box without the nodes gives
  nulls, never an error."""

  def __init__(self, index: int):
    import glob
    import threading

    self._threading = threading
    self.samples: list[tuple[float, float]] = []
    self.props = torch.cuda.get_device_properties(index)
    self.node = None
    cands = []
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
      try:
        if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
          continue
        hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
        if hw and os.path.exists(os.path.join(hw[0], "freq1_input")):
          slot = ""
          for ln in open(os.path.join(dev, "uevent")):
            if ln.startswith("PCI_SLOT_NAME="):
              slot = ln.split("=", 1)[1].strip().lower()
          cands.append((slot, hw[0]))
      except OSError:
        continue
    want = ""
    try:
      want = f"{self.props.pci_domain_id:04x}:{self.props.pci_bus_id:02x}:{self.props.pci_device_id:02x}.0"
    except AttributeError:
      pass
    for slot, hw in cands:
      if slot == want:
        self.node = hw
    if self.node is None and len(cands) == 1:
      self.node = cands[0][1]
    self._power = None
    if self.node:
      for f in ("power1_average", "power1_input"):
        if os.path.exists(os.path.join(self.node, f)):
          self._power = os.path.join(self.node, f)
          break
    self._stop = None
    self._thread = None

  def _read(self):
    try:
      mhz = int(open(os.path.join(self.node, "freq1_input")).read()) / 1e6
      watts = int(open(self._power).read()) / 1e6 if self._power else float("nan")
      return mhz, watts
    except (OSError, ValueError):
      return None

  def start(self, period_s: float = 0.010) -> None:
    """One sample now, then one every ``period_s`` (and a last one at stop()).  10 ms: `value` is wall-clock based and includes host launch time, and a
    Python thread that wakes every millisecond contends for the GIL with the launch loop of a launch-bound workload (decode: ~ 100 us per step) — at
    10 ms the sampler takes the GIL a handful of times per timed region; the round-4 lines were taken at 1 ms."""
    if not self.node:
      return
    self.samples = []
    self._stop = self._threading.Event()

    def loop():
      while True:
        r = self._read()
        if r:
          self.samples.append(r)
        if self._stop.wait(period_s):
          break
      r = self._read()
      if r:
        self.samples.append(r)

    self._thread = self._threading.Thread(target=loop, daemon=True)
    self._thread.start()

  def stop(self) -> None:
    if self._thread is not None:
      self._stop.set()
      self._thread.join()
      self._thread = None

  def summary(self) -> dict:
    cus = self.props.multi_processor_count
    fmax = getattr(self.props, "clock_rate", 0) / 1e6  # kHz -> GHz (torch 2.10 + ROCm 7 does not fill it in: then the top DPM level of sysfs)
    if not fmax and self.node:
      try:
        import re

        levels = [int(m.group(1)) for m in re.finditer(r"(\d+)\s*[Mm][Hh]z", open(os.path.join(os.path.dirname(os.path.dirname(self.node)), "pp_dpm_sclk")).read())]
        fmax = max(levels) / 1e3 if levels else 0.0
      except OSError:
        fmax = 0.0
    out = {"name": self.props.name, "cus": cus, "max_sclk_mhz": round(fmax * 1e3, 1) if fmax else None,
           # the formula of SURVEY.md section 8d: CUs x 4 SIMDs x 1024 bf16 MFMA FLOP / clk x f_max
           "peak_tflops_from_device": round(cus * 4 * 1024 * fmax / 1e3, 1) if fmax else None,
           "telemetry_source": os.path.relpath(self.node, "/sys/class/drm") if self.node else None}
    if self.samples:
      f = [x[0] for x in self.samples]
      p = [x[1] for x in self.samples if x[1] == x[1]]
      out.update(sclk_mhz_avg=round(sum(f) / len(f), 1), sclk_mhz_min=round(min(f), 1), sclk_mhz_max=round(max(f), 1),
                 power_w_avg=round(sum(p) / len(p), 1) if p else None, power_w_max=round(max(p), 1) if p else None, samples=len(f),
                 sampled="freq1_input / power1_* of the GPU's hwmon node, at its start, every 10 ms and at its end, over the timed region")
    return out


def cpu_baseline(w: dict, seconds_budget: float = 25.0) -> dict:
  """The reference's CPU path (torch CPU SDPA; ffpa_attn_func falls back to it) on this box's host cores.  A small sample
  (at most 4 query heads of one batch element) is timed first; if the whole workload then fits the budget (about 10-30 s
  of CPU work) it is run in full — config 2 takes about 4.5 s per pass on the pool's 256 host threads — else the sample
  is the reported figure."""
  cores = os.cpu_count() or 1
  torch.set_num_threads(cores)
  g = w["Hq"] // w["Hkv"]
  Nq, Nkv, D = w["Nq"], w["Nkv"], w["D"]
  kw = {}
  if w["mask"] == "tril_bool" or w["via"] == "op_offset0" or w["causal"]:
    kw["is_causal"] = True  # top-left (what the mask / causal_offset=0 express), or Nq == Nkv where both alignments coincide
  elif w["mask"] == "key_bias":
    kw["attn_mask"] = make_mask(w, torch.bfloat16, "cpu")

  def run(B, Hq, Hkv, reps):
    torch.manual_seed(0)
    q = torch.randn(B, Hq, Nq, D, dtype=torch.bfloat16)
    k = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16)
    v = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16)
    torch._C._nn.scaled_dot_product_attention(q, k, v, enable_gqa=Hq != Hkv, **kw)  # warm-up
    best = float("inf")
    for _ in range(reps):
      t0 = time.perf_counter()
      torch._C._nn.scaled_dot_product_attention(q, k, v, enable_gqa=Hq != Hkv, **kw)
      best = min(best, time.perf_counter() - t0)
    return best, (q, k, v)

  sHq = min(4, w["Hq"]) if g == 1 else g  # keep whole GQA groups
  sHkv = max(1, sHq // g)
  t_begin = time.perf_counter()
  best, (q, k, v) = run(1, sHq, sHkv, 3)
  B, Hq, Hkv, reps, what = 1, sHq, sHkv, 3, "a sample"
  full_estimate = best * w["B"] * w["Hq"] / sHq
  if (w["B"], w["Hq"]) != (1, sHq) and full_estimate * 3.2 < seconds_budget - (time.perf_counter() - t_begin):
    best, _ = run(w["B"], w["Hq"], w["Hkv"], 2)
    B, Hq, Hkv, reps, what = w["B"], w["Hq"], w["Hkv"], 2, "the whole workload"
  flops = valid_pairs_flops(dict(w, Hq=Hq, Hkv=Hkv), B)
  out = {
    "value": round(flops / best / 1e12, 4),
    "unit": "TFLOPS",
    "cores": torch.get_num_threads(),
    "kind": "reference",
    "sample": f"torch CPU SDPA (the reference's CPU path: ffpa_attn_func falls back to it) on {what}: B={B} Hq={Hq} Hkv={Hkv} Nq={Nq} Nkv={Nkv} "
              f"D={D} bf16{' causal' if 'is_causal' in kw else ''}{' + key bias' if 'attn_mask' in kw else ''}, best of {reps} after 1 warm-up, "
              f"{best * 1e3:.1f} ms per pass",
  }
  try:  # the oracle (scalar C restatement of the kernel's recurrence), for scale: one thread, <= 64 rows of one head
    from oracle import ffpa_oracle as fo
    rows = min(64, Nq)
    qb, dname = fo.torch_to_bits(q[:, :1, :rows].contiguous())
    kb, _ = fo.torch_to_bits(k[:, :1].contiguous())
    vb, _ = fo.torch_to_bits(v[:, :1].contiguous())
    t0 = time.perf_counter()
    fo.oracle_forward(qb, kb, vb, dname, scale=D ** -0.5)
    dt = time.perf_counter() - t0
    out["oracle_port"] = {"value": round(attention_fwd_flops(1, 1, rows, Nkv, D) / dt / 1e12, 6), "unit": "TFLOPS", "cores": 1,
                          "kind": "port", "sample": f"oracle/ffpa_oracle.c on {rows} rows x {Nkv} keys of one head, D={D}, {dt * 1e3:.0f} ms"}
  except Exception as exc:  # the oracle is test infrastructure: its absence must not break the bench line
    out["oracle_port"] = {"error": str(exc)[:120]}
  return out


def accuracy(w: dict, q, k, v, mask, scale: float) -> dict:
  """max / mean |O - O_sdpa| on the whole tensors (fp32 promote, cli/_runner_fwd.py:178-200), max |LSE - LSE_ref| with
  LSE_ref = fp32 logsumexp of the scores of the first and last head of batch 0, and the same-device SDPA timing."""
  from ffpa_attn_amd import hip

  res = {}
  gqa = w["Hq"] != w["Hkv"]
  top_left = w["mask"] == "tril_bool" or w["via"] == "op_offset0"
  sdpa_kw = dict(enable_gqa=gqa)
  if top_left or w["causal"]:
    sdpa_kw["is_causal"] = True
  elif mask is not None:
    sdpa_kw["attn_mask"] = mask
  bias = None if (w["via"] == "op_offset0") else mask
  o, lse = hip.ffpa_attn_forward_hip(q, k, v, bias if bias is None or bias.dim() == 4 else bias.view(1, 1, *bias.shape), causal=bool(w["causal"]),
                                     softmax_scale=scale, causal_offset=0 if w["via"] == "op_offset0" else None)
  ref = torch.nn.functional.scaled_dot_product_attention(q, k, v, **sdpa_kw)
  d = (o.float() - ref.float()).abs()
  res["max_abs_err_vs_sdpa"] = round(d.max().item(), 6)
  res["mean_abs_err_vs_sdpa"] = round(d.mean().item(), 8)
  # the same two outputs against fp32 MATH (softmax(scale*QK^T [+mask]) V in fp32, heads {0, Hq-1} of batch 0): which side of a
  # kernel-vs-SDPA difference is off.  Both store bf16, so neither can beat half an output ulp (2^-9 |O|).
  lse_err = err_math = err_math_sdpa = 0.0
  g = w["Hq"] // w["Hkv"]
  rows = torch.arange(w["Nq"], device=q.device)[:, None]
  cols = torch.arange(w["Nkv"], device=q.device)[None, :]
  for h in sorted({0, w["Hq"] - 1}):
    s = (q[0, h].float() @ k[0, h // g].float().T) * scale
    if top_left:
      s = s.masked_fill(cols > rows, float("-inf"))
    elif w["causal"]:
      s = s.masked_fill(cols > rows + (w["Nkv"] - w["Nq"]), float("-inf"))
    elif mask is not None:
      s = s + mask.float().reshape(-1, w["Nkv"])  # the bench's additive masks are key biases: [1, Nkv] broadcasts over the rows
    lse_err = max(lse_err, (lse[0, h] - torch.logsumexp(s, -1)).abs().max().item())
    want = torch.softmax(s, -1) @ v[0, h // g].float()
    err_math = max(err_math, (o[0, h].float() - want).abs().max().item())
    err_math_sdpa = max(err_math_sdpa, (ref[0, h].float() - want).abs().max().item())
    del s, want
  res["max_abs_lse_err"] = round(lse_err, 7)
  res["max_abs_err_vs_fp32_math"] = round(err_math, 6)
  res["sdpa_max_abs_err_vs_fp32_math"] = round(err_math_sdpa, 6)
  res["lse_ref"] = "fp32 logsumexp(scale*QK^T [+mask]) of heads {0, Hq-1} of batch 0 (the fp32-math errors use the same heads)"
  flops = valid_pairs_flops(w, w["B"])
  for _ in range(2):
    torch.nn.functional.scaled_dot_product_attention(q, k, v, **sdpa_kw)
  torch.cuda.synchronize()
  reps = 5
  t1 = time.perf_counter()
  for _ in range(reps):
    torch.nn.functional.scaled_dot_product_attention(q, k, v, **sdpa_kw)
  torch.cuda.synchronize()
  sdpa_s = (time.perf_counter() - t1) / reps
  res["sdpa_gpu_tflops"] = round(flops / sdpa_s / 1e12, 2)
  res["sdpa_gpu_ms"] = round(sdpa_s * 1e3, 4)
  res["sdpa_call"] = "F.scaled_dot_product_attention(" + ", ".join(f"{a}={'<mask>' if a == 'attn_mask' else b}" for a, b in sdpa_kw.items()) + ")"
  return res


def varlen_decode_main(args) -> None:
  """`--workload varlen_decode` (N = 1): a packed DECODE batch — one query token per sequence, ragged KV lengths, GQA — through the packed-sequence op
  (hip.varlen_forward = ffpa_attn::_varlen_fwd_hip -> ffpa_attn_varlen_fwd: ffpa_fwd_m16_varlen_kernel with the heads of a KV group in the rows of one tile and
  every sequence's KV range split by its own length + ffpa_varlen_merge_kernel).  Same contract as the dense `decode` workload: W warm-ups, exactly K timed
  steps between synchronises, HIP events per step; the roofline is HBM's (every K / V byte once: algorithmic bytes / step time).  Beside it: the same call with one
  KV range per sequence, with one workgroup per QUERY head, and the batch as a loop of dense decode calls."""
  import numpy as np

  from ffpa_attn_amd import hip

  name = "varlen_decode"
  w = WORKLOADS[name]
  nseq, Hq, Hkv, D = w["B"], w["Hq"], w["Hkv"], w["D"]
  dev = torch.device("cuda", 0)
  torch.cuda.set_device(dev)
  hip.load_library()
  lens = [int(x) for x in np.random.default_rng(0).integers(w["kv_range"][0], w["kv_range"][1], size=nseq)]
  tk = sum(lens)
  torch.manual_seed(0)
  q = torch.randn(nseq, Hq, D, dtype=torch.bfloat16, device=dev)
  k = torch.randn(tk, Hkv, D, dtype=torch.bfloat16, device=dev)
  v = torch.randn(tk, Hkv, D, dtype=torch.bfloat16, device=dev)
  cu_q = torch.arange(0, nseq + 1, dtype=torch.int32, device=dev)
  bk = [0]
  for n in lens:
    bk.append(bk[-1] + n)
  cu_k = torch.tensor(bk, dtype=torch.int32, device=dev)
  scale = D ** -0.5
  flops = 4 * Hq * D * tk  # every token sees every key of its sequence
  alg_bytes = 2 * D * (2 * tk * Hkv + 2 * nseq * Hq) + 4 * Hq * nseq  # K + V once, q, o, LSE

  def step(**kw):
    return hip.varlen_forward(q, k, v, cu_q, cu_k, 1, max(lens), True, scale, **kw)

  telemetry = DeviceTelemetry(0)

  def timed(fn, telem=False):
    for _ in range(args.warmup):
      fn()
    torch.cuda.synchronize()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    if telem:
      telemetry.start()
    t0 = time.perf_counter()
    for i in range(args.steps):
      starts[i].record()
      fn()
      ends[i].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if telem:
      telemetry.stop()
    return elapsed, sorted(a.elapsed_time(b) for a, b in zip(starts, ends))

  elapsed, kernel_ms = timed(step, telem=True)
  device = telemetry.summary()
  kernel_ms_avg = sum(kernel_ms) / len(kernel_ms)
  plan = {}
  out, lse = step(plan_out=plan)
  steady = None
  if not args.no_steady:
    n_pre, n_timed = max(20, int(150.0 / kernel_ms_avg)), max(20, min(400, int(60.0 / kernel_ms_avg)))
    for _ in range(n_pre):
      step()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(n_timed):
      step()
    ev1.record()
    torch.cuda.synchronize()
    ss_ms = ev0.elapsed_time(ev1) / n_timed
    steady = {"ms_per_step": round(ss_ms, 4), "gbps": round(alg_bytes / ss_ms / 1e6, 1), "frac_of_hbm_peak": round(alg_bytes / ss_ms / 1e6 / HBM_PEAK_GBPS, 4), "launches": n_timed,
              "after_launches": n_pre, "what": "the same step back to back after >= 150 ms of continuous load, one HIP event pair around the launches; outside the timed region"}
  build = build_identity()
  traffic, traffic_src, traffic_stale = measured_traffic(name, build.get("lib_sha16"))
  note = None
  under_profiler = any(k_.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k_ in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "").lower()
  if not args.no_live_traffic and not under_profiler:
    torch.cuda.synchronize()
    live, why = live_traffic(name)
    if live is not None:
      traffic, traffic_src, traffic_stale = live, why, False
    else:
      note = why
  achieved = alg_bytes / (kernel_ms_avg * 1e-3) / 1e9
  roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
          "traffic_source": traffic_src, "kernel": plan["kernel"], "kernel_ms_avg": round(kernel_ms_avg, 4), "kernel_ms_median": round(kernel_ms[len(kernel_ms) // 2], 4),
          "bytes_per_launch": alg_bytes, "algorithmic_bytes_per_launch": alg_bytes, "workgroups": plan["workgroups"], "traffic_stale": bool(traffic_stale),
          "what": "HIP events around one step (the split kernel + its merge kernel); traffic: the split kernel's dispatches"}
  if note is not None:
    roof["traffic_live_failed"] = note
  line = {
    "metric": f"attention fwd TFLOPS + max-abs-err vs SDPA, bf16 packed decode batch {nseq} x 1 token, KV {min(lens)} ... {max(lens)} (sum {tk}) Hq={Hq}/Hkv={Hkv} D={D} [varlen_decode]",
    "value": round(flops * args.steps / elapsed / 1e12, 3), "unit": "TFLOPS", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
    "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
    "config": {"workload": f"varlen_decode: {w['note']}; T_k={tk} Hq={Hq} Hkv={Hkv} D={D} bf16", "global_batch": nseq, "seq_len": 1, "parallelism": "single GPU",
               "flops_model": "4*Hq*D*sum_i nkv_i", "step": "hip.varlen_forward (ffpa_attn::_varlen_fwd_hip)"},
    "roofline": roof, "device": device, "steady_state": steady, "build": build,
    "plan": {k_: plan[k_] for k_ in ("row_tiles", "block_rows", "block_keys", "workgroups", "splits")},
  }
  if not args.no_sdpa:  # (the comparison legs; --no-sdpa: none of them — the profiler passes)
    legs = {}
    for what, kw in (("one_kv_range", dict(num_splits=1)), ("one_workgroup_per_query_head", dict(flags=hip.FLAG_NO_PACK_GQA)), ("no_nt_hint", dict(flags=hip.FLAG_NO_KV_STREAM))):
      el, _ = timed(lambda kw=kw: step(**kw))
      pl = {}
      step(plan_out=pl, **kw)
      legs[what] = {"ms_per_step": round(el / args.steps * 1e3, 4), "gbps": round(alg_bytes * args.steps / el / 1e9, 1), "kernel": pl["kernel"], "splits": pl["splits"], "workgroups": pl["workgroups"],
                    "library_speedup": round(el / elapsed, 3)}
    line["other_launches"] = legs

    def seq(t, a, b):
      return t[a:b].transpose(0, 1).unsqueeze(0)

    def loop():
      return [hip.ffpa_attn_forward_hip(seq(q, i, i + 1), seq(k, bk[i], bk[i + 1]), seq(v, bk[i], bk[i + 1]), None, causal=True, softmax_scale=scale)[0] for i in range(nseq)]

    el, _ = timed(loop)
    ref = loop()
    diff = max((out[i:i + 1].float() - ref[i][0].transpose(0, 1).float()).abs().max().item() for i in range(nseq))
    line["per_sequence_loop"] = {"ms_per_step": round(el / args.steps * 1e3, 4), "gbps": round(alg_bytes * args.steps / el / 1e9, 1), "launches_per_step": 2 * nseq,
                                 "max_abs_diff_vs_packed": round(diff, 6), "packed_speedup": round(el / elapsed, 3),
                                 "what": "the dense decode call (split-KV kernel + merge) per sequence on zero-copy views of the packed tensors; same W / K"}
    try:
      sd = lambda: [torch.nn.functional.scaled_dot_product_attention(seq(q, i, i + 1), seq(k, bk[i], bk[i + 1]), seq(v, bk[i], bk[i + 1]), enable_gqa=True) for i in range(nseq)]  # noqa: E731
      ref = sd()
      err = max((out[i:i + 1].float() - ref[i][0].transpose(0, 1).float()).abs().max().item() for i in range(nseq))
      torch.cuda.synchronize()
      t1 = time.perf_counter()
      for _ in range(3):
        sd()
      torch.cuda.synchronize()
      sdpa_s = (time.perf_counter() - t1) / 3
      line.update(max_abs_err_vs_sdpa=round(err, 6), sdpa_gpu_ms=round(sdpa_s * 1e3, 4), speedup_vs_sdpa_gpu=round(sdpa_s / (elapsed / args.steps), 3),
                  sdpa_call="F.scaled_dot_product_attention(enable_gqa=True) per sequence")
    except Exception as e:  # noqa: BLE001 — informative only
      line["sdpa_error"] = str(e)[:200]
  print(json.dumps(line))


def varlen_main(args) -> None:
  """`--workload varlen` (N = 1): the packed-sequence call — ffpa_attn_varlen_func -> ffpa_attn::_varlen_fwd_hip -> ffpa_attn_varlen_fwd -> ONE launch of
  ffpa_fwd_m16_varlen_kernel — under the same contract as the dense workloads: W warm-ups, exactly K timed steps between synchronises, HIP events per
  step, the MFMA roofline of the kernel with its HBM traffic measured in the run, the reference's CPU path beside it.  Next to the contract figure: the
  same batch as a per-sequence loop of dense calls (what the reference tells callers without its CuTe-DSL backend to do) and SDPA per sequence."""
  import math

  from ffpa_attn_amd import ffpa_attn_varlen_func, hip

  name = "varlen"
  w = WORKLOADS[name]
  lens, Hq, Hkv, D = list(w["lens"]), w["Hq"], w["Hkv"], w["D"]
  dev = torch.device("cuda", 0)
  torch.cuda.set_device(dev)
  hip.load_library()
  total = sum(lens)
  torch.manual_seed(0)
  q = torch.randn(total, Hq, D, dtype=torch.bfloat16, device=dev)
  k = torch.randn(total, Hkv, D, dtype=torch.bfloat16, device=dev)
  v = torch.randn(total, Hkv, D, dtype=torch.bfloat16, device=dev)
  bounds = [0]
  for n in lens:
    bounds.append(bounds[-1] + n)
  cu = torch.tensor(bounds, dtype=torch.int32, device=dev)
  max_len = max(lens)
  flops = 4 * Hq * D * sum(n * (n + 1) // 2 for n in lens)  # the reference's model (cli/_flops.py:37-53) per sequence: causal self-attention sees n (n + 1) / 2 pairs
  alg_bytes = 2 * D * total * (2 * Hq + 2 * Hkv) + 4 * Hq * total

  def step():
    return ffpa_attn_varlen_func(q, k, v, cu, cu, max_len, max_len, causal=True, enable_gqa=True)

  def seq(t, i):
    return t[bounds[i]:bounds[i + 1]].transpose(0, 1).unsqueeze(0)  # [n, H, D] -> [1, H, n, D], a view

  def step_loop():  # the same batch, one dense call per sequence (zero-copy views: the C-ABI honours strides)
    return [ffpa_attn_func(seq(q, i), seq(k, i), seq(v, i), is_causal=True, enable_gqa=True) for i in range(len(lens))]

  telemetry = DeviceTelemetry(0)

  def timed(fn):
    for _ in range(args.warmup):
      fn()
    torch.cuda.synchronize()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    telemetry.start()
    t0 = time.perf_counter()
    for i in range(args.steps):
      starts[i].record()
      fn()
      ends[i].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    telemetry.stop()
    return elapsed, sorted(a.elapsed_time(b) for a, b in zip(starts, ends))

  elapsed, kernel_ms = timed(step)
  device = telemetry.summary()
  kernel_ms_avg = sum(kernel_ms) / len(kernel_ms)
  value = flops * args.steps / elapsed / 1e12
  steady = None
  if not args.no_steady:
    n_pre, n_timed = max(20, int(150.0 / kernel_ms_avg)), max(20, min(400, int(60.0 / kernel_ms_avg)))
    for _ in range(n_pre):
      step()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(n_timed):
      step()
    ev1.record()
    torch.cuda.synchronize()
    ss_ms = ev0.elapsed_time(ev1) / n_timed
    steady = {"ms_per_step": round(ss_ms, 4), "tflops": round(flops / ss_ms / 1e9, 2), "launches": n_timed, "after_launches": n_pre,
              "what": "the same step back to back after >= 150 ms of continuous load, one HIP event pair around the launches; outside the timed region"}
  plan = hip.varlen_launch_plan(len(lens), Hq, Hkv, max_len, max_len, D, causal=True)
  build = build_identity()
  traffic, traffic_src, traffic_stale = measured_traffic(name, build.get("lib_sha16"))
  note = None
  under_profiler = any(k_.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k_ in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "").lower()
  if not args.no_live_traffic and not under_profiler:
    torch.cuda.synchronize()
    live, why = live_traffic(name)
    if live is not None:
      traffic, traffic_src, traffic_stale = live, why, False
    else:
      note = why
  achieved = flops / (kernel_ms_avg * 1e-3) / 1e12
  roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4),
          "traffic": traffic, "traffic_source": traffic_src, "kernel": plan["kernel"], "kernel_ms_avg": round(kernel_ms_avg, 4),
          "kernel_ms_median": round(kernel_ms[len(kernel_ms) // 2], 4), "flops_per_launch": flops, "algorithmic_bytes_per_launch": alg_bytes,
          "workgroups": plan["workgroups"], "traffic_stale": bool(traffic_stale)}
  if note is not None:
    roof["traffic_live_failed"] = note
  if device.get("peak_tflops_from_device"):
    roof["frac_of_device_peak"] = round(achieved / device["peak_tflops_from_device"], 4)
  line = {
    "metric": f"attention fwd TFLOPS + max-abs-err vs SDPA, bf16 packed sequences {lens} Hq={Hq}/Hkv={Hkv} D={D} causal [varlen]",
    "value": round(value, 2), "unit": "TFLOPS", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
    "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
    "config": {"workload": f"varlen: {w['note']}; T={total} Hq={Hq} Hkv={Hkv} D={D} bf16", "global_batch": len(lens), "seq_len": max_len, "parallelism": "single GPU",
               "flops_model": "4*Hq*D*sum_i n_i (n_i + 1) / 2", "step": "ffpa_attn_varlen_func"},
    "roofline": roof, "device": device, "steady_state": steady, "build": build,
    "plan": {k_: plan[k_] for k_ in ("row_tiles", "block_rows", "block_keys", "workgroups")},
  }
  # the same batch as a loop of dense calls, one per sequence — and its outputs against the packed call's (same tile, same recurrence: the bits where the
  # dense plan is the plain tile; rounding where it pairs / splits / takes another tile)
  out = step()
  if not args.no_sdpa:  # (--no-sdpa: no comparison legs at all — what the profiler passes of tools/gpu_round.sh run, so that their kernel list is this kernel's alone)
    loop_elapsed, _ = timed(step_loop)
    loop_out = step_loop()
    diff = max((out[bounds[i]:bounds[i + 1]].float() - loop_out[i][0].transpose(0, 1).float()).abs().max().item() for i in range(len(lens)))
    line["per_sequence_loop"] = {"ms_per_step": round(loop_elapsed / args.steps * 1e3, 4), "tflops": round(flops * args.steps / loop_elapsed / 1e12, 2), "launches_per_step": len(lens),
                                 "max_abs_diff_vs_packed": round(diff, 6), "packed_speedup": round(loop_elapsed / elapsed, 3),
                                 "what": "ffpa_attn_func(is_causal=True) per sequence on zero-copy views of the packed tensors (each call its own launch plan); same W / K"}
    try:
      sd = lambda: [torch.nn.functional.scaled_dot_product_attention(seq(q, i), seq(k, i), seq(v, i), is_causal=True, enable_gqa=True) for i in range(len(lens))]  # noqa: E731
      ref = sd()
      err = max((out[bounds[i]:bounds[i + 1]].float() - ref[i][0].transpose(0, 1).float()).abs().max().item() for i in range(len(lens)))
      torch.cuda.synchronize()
      t1 = time.perf_counter()
      for _ in range(3):
        sd()
      torch.cuda.synchronize()
      sdpa_s = (time.perf_counter() - t1) / 3
      line.update(max_abs_err_vs_sdpa=round(err, 6), sdpa_gpu_ms=round(sdpa_s * 1e3, 4), sdpa_gpu_tflops=round(flops / sdpa_s / 1e12, 2),
                  speedup_vs_sdpa_gpu=round(sdpa_s / (elapsed / args.steps), 3), sdpa_call="F.scaled_dot_product_attention(is_causal=True, enable_gqa=True) per sequence")
    except Exception as e:  # noqa: BLE001 — informative only
      line["sdpa_error"] = str(e)[:200]
  if not args.no_cpu_baseline:
    # the reference's CPU path for a packed batch: it has none (the CuTe-DSL backend is GPU-only) — what it tells such callers to do is the dense call per
    # sequence, which falls back to torch CPU SDPA: timed on the THREE shortest sequences (a bounded sample)
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    sample = sorted(range(len(lens)), key=lambda i: lens[i])[:3]
    cq, ck, cv = q.cpu(), k.cpu(), v.cpu()
    run = lambda: [torch._C._nn.scaled_dot_product_attention(seq(cq, i), seq(ck, i), seq(cv, i), is_causal=True, enable_gqa=True) for i in sample]  # noqa: E731
    run()
    best = float("inf")
    for _ in range(2):
      t0 = time.perf_counter()
      run()
      best = min(best, time.perf_counter() - t0)
    sflops = 4 * Hq * D * sum(lens[i] * (lens[i] + 1) // 2 for i in sample)
    line["cpu_baseline"] = {"value": round(sflops / best / 1e12, 4), "unit": "TFLOPS", "cores": torch.get_num_threads(), "kind": "reference",
                            "sample": f"torch CPU SDPA per sequence (what the reference's dense entry point falls back to; its packed entry point has no CPU path) on the "
                                      f"three shortest sequences {[lens[i] for i in sample]}, Hq={Hq} Hkv={Hkv} D={D} bf16 causal, best of 2 after 1 warm-up, {best * 1e3:.1f} ms per pass"}
  print(json.dumps(line), flush=True)


def stub_main(args, world: int, rank: int) -> None:
  """(tests) The launch / barrier / max-over-ranks / one-JSON-line plumbing of this script on CPU over `--stub-backend` (gloo), with a
  short sleep standing in for the step: what tests/test_bench_spawn.py runs at world size 2.  Not a benchmark."""
  import torch.distributed as dist

  if world > 1:
    dist.init_process_group(args.stub_backend)
  for _ in range(args.warmup):
    time.sleep(0.001)
  if world > 1:
    dist.barrier()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    time.sleep(0.002 * (1 + rank))  # ranks differ: the reported time must be the slowest rank's
  local = time.perf_counter() - t0
  if world > 1:
    dist.barrier()
  total = time.perf_counter() - t0
  every = [local]
  if world > 1:
    t = torch.tensor([total], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    total = float(t.item())
    mine = torch.tensor([local], dtype=torch.float64)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    every = [float(x.item()) for x in got]
  with_gather = None
  if world > 1:
    # the gather leg's plumbing with a stand-in for the kernel: the REAL sharding.attend_and_gather_units (transport probe, agreed piece count, point-to-point
    # sends into the final slices) on small CPU tensors, and the report the line carries about it
    from ffpa_attn_amd import sharding

    n_units, grp, nq, nkv, d = 2 * world, 2, 8, 16, 16
    u0, u1 = sharding.partition_units(n_units, world, rank)
    q, k, v = sharding.synthetic_unit_block(u0, u1, grp, nq, nkv, d, dtype=torch.float32, device="cpu", seed=0)
    real = sharding.attend_units
    sharding.attend_units = lambda a, b, c, **kw: torch.nn.functional.scaled_dot_product_attention(a, b, c, enable_gqa=True)
    try:
      stats = {}
      t1 = time.perf_counter()
      full = sharding.attend_and_gather_units(q, k, v, n_units, chunks=args.gather_chunks, stats=stats)
      g_ms = (time.perf_counter() - t1) * 1e3
      o_local = sharding.attend_units(q, k, v)
      t1 = time.perf_counter()
      alone = sharding.gather_units(o_local, n_units)
      p_ms = (time.perf_counter() - t1) * 1e3
    finally:
      sharding.attend_units = real
    ok = bool(torch.equal(full, alone))
    with_gather = {**gather_report(stats, o_local.numel() * o_local.element_size(), world, 0.0, g_ms, p_ms, args.gather_chunks), "equal_to_plain_gather": ok}
  if rank == 0:
    print(json.dumps({"stub": True, "backend": args.stub_backend, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": round(total / args.steps * 1e3, 4), "per_rank_s": [round(x, 4) for x in every],
                      **({"with_gather": with_gather} if with_gather is not None else {})}), flush=True)
  if world > 1:
    dist.destroy_process_group()


def sweep_main(args) -> None:
  """The reference bench's one-shot sweep (`python -m ffpa_attn.bench`: cli/_runner_fwd.py:599-672 runs self / cross / decode / gqa /
  causal / attn-mask / dropout / non-aligned for a head dim in one process and prints one table, :473): the same eight cases at its
  defaults (B 1, H 32, N 8192) for every head dim of --sweep-dims, this kernel next to SDPA on the same GPU, one table per head dim on
  stderr and ONE JSON line on stdout."""
  from ffpa_attn_amd import hip

  dev = torch.device("cuda", 0)
  hip.load_library()
  B, H, N = 1, 32, 8192
  out = {}
  for D in [int(x) for x in args.sweep_dims.split(",") if x]:
    cases = {
      "self": _w(B, H, H, N, N, D), "cross": _w(B, H, H, 1024, N, D), "decode": _w(B, H, H, 1, N, D, bound="hbm"), "gqa": _w(B, H, H // 4, N, N, D),
      "causal": _w(B, H, H, N, N, D, causal=True), "attn_mask": _w(B, H, H, N, N, D, mask="key_bias"), "dropout": _w(B, H, H, N, N, D, dropout=0.1),
      "non_aligned": _w(B, H // 4, H // 4, N - 1, N - 1, D),
    }
    rows = []
    for cname, w in cases.items():
      torch.manual_seed(0)
      q = torch.randn(w["B"], w["Hq"], w["Nq"], D, dtype=torch.bfloat16, device=dev)
      k = torch.randn(w["B"], w["Hkv"], w["Nkv"], D, dtype=torch.bfloat16, device=dev)
      v = torch.randn(w["B"], w["Hkv"], w["Nkv"], D, dtype=torch.bfloat16, device=dev)
      mask = make_mask(w, torch.bfloat16, dev)
      kw = dict(attn_mask=mask, dropout_p=w["dropout"], is_causal=w["causal"], enable_gqa=w["Hq"] != w["Hkv"])

      def timeit(fn, reps):
        for _ in range(2):
          fn()
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
          fn()
        b_.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b_) / reps

      ms = timeit(lambda: ffpa_attn_func(q, k, v, **kw), args.steps)
      flops = valid_pairs_flops(w, w["B"])
      rec = {"ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 1)}
      if w["bound"] == "hbm":
        rec["gbps"] = round(algorithmic_bytes(w, w["B"]) / ms / 1e6, 1)
      if not w["dropout"]:
        sdpa = lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=mask, is_causal=w["causal"], enable_gqa=w["Hq"] != w["Hkv"])  # noqa: E731
        try:
          sms = timeit(sdpa, 3)
          rec["sdpa_ms"], rec["speedup"] = round(sms, 4), round(sms / ms, 2)
          rec["max_abs_err_vs_sdpa"] = round((ffpa_attn_func(q, k, v, **kw).float() - sdpa().float()).abs().max().item(), 6)
        except Exception as e:  # noqa: BLE001
          rec["sdpa_error"] = str(e)[:80]
      rec["kernel"] = planned_kernel(w, q, k, v, mask, D ** -0.5).get("kernel")
      rows.append((cname, rec))
      del q, k, v, mask
    out[str(D)] = dict(rows)
    print(f"\n== D = {D}  (B {B}, H {H}, N {N}, bf16; bench.py --sweep) ==", file=sys.stderr)
    print(f"{'case':<12} {'ms':>9} {'TFLOPS':>8} {'SDPA ms':>9} {'speedup':>8} {'max|err|':>9}  kernel", file=sys.stderr)
    for cname, r in rows:
      print(f"{cname:<12} {r['ms']:>9.4f} {r['tflops']:>8.1f} {r.get('sdpa_ms', float('nan')):>9.4f} {r.get('speedup', float('nan')):>8.2f} "
            f"{r.get('max_abs_err_vs_sdpa', float('nan')):>9.2e}  {r['kernel']}", file=sys.stderr)
  print(json.dumps({"sweep": out, "shape": f"B={B} H={H} N={N} bf16", "steps": args.steps, "build": build_identity(),
                    "cases": "self / cross (Nq 1024) / decode (Nq 1) / gqa (H/4) / causal / attn_mask ([1,1,1,Nkv]*0.25) / dropout 0.1 / non_aligned (N-1, H/4): cli/_runner_fwd.py:599-672"}),
        flush=True)


def free_port() -> int:
  with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
    sk.bind(("127.0.0.1", 0))
    return sk.getsockname()[1]


def spawn_ranks(n: int, argv: list[str]) -> int:
  """`python bench.py --gpus N` without a launcher: start N copies of this script, one per GPU, with the torchrun environment
  (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT).  Rank 0 keeps this process's stdout (its one JSON
  line is the bench line), the other ranks' stdout is dropped, every rank's stderr is passed through.  Returns the worst exit code."""
  port = free_port()
  procs = []
  for r in range(n):
    env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env, stdout=None if r == 0 else subprocess.DEVNULL))
  worst = 0
  try:
    live = list(procs)
    while live:  # poll ALL ranks: a rank that dies leaves the others waiting in a rendezvous / collective for minutes
      time.sleep(0.2)
      for pr in list(live):
        rc = pr.poll()
        if rc is None:
          continue
        live.remove(pr)
        worst = worst or rc
        if rc != 0:
          for other in live:
            other.terminate()
  finally:
    for pr in procs:
      if pr.poll() is None:
        pr.kill()
  return worst


XGMI_LINK_GBPS = 153.0  # one xGMI link, one direction (7 links per GPU, a full mesh inside the node)


def gather_report(stats: dict, shard_bytes: int, world: int, step_ms: float, with_ms: float, pure_ms: "float | None", chunks_asked: int) -> dict:
  """What the line says about the gather of O besides its throughput: which transport ran (sharding.gather_transport: RCCL point-to-point, or the
  all_gather_into_tensor fallback), into how many pieces the ranks agreed to cut a block, and the figures that let a first multi-GPU run judge itself —
  every rank receives world - 1 shards, each over its own link: shard_bytes / 153 GB/s is what a gather that uses the mesh perfectly costs
  (256 MiB: 1.75 ms), `exposed_ms` is what the overlapped form adds to the kernel-only step, `alone_ms` is the gather with nothing to hide behind."""
  expected = shard_bytes / (XGMI_LINK_GBPS * 1e9) * 1e3 if world > 1 else 0.0
  what = {"p2p": "each piece sent point-to-point (RCCL batch_isend_irecv) into its final slice on every other rank while the next piece computes",
          "all_gather": "RCCL point-to-point unavailable on this node (probe): each piece all_gather_into_tensor'ed into a world x piece temporary while the next piece computes, then copied to its slices",
          "local": "one rank: a device-to-device copy"}.get(stats.get("transport"), str(stats.get("transport")))
  rep = {"transport": stats.get("transport"), "chunks": stats.get("chunks"), "chunks_requested": chunks_asked,
         "shard_bytes": int(shard_bytes), "received_bytes_per_rank": int(shard_bytes) * (world - 1),
         "expected_ms_one_shard_per_link": round(expected, 6), "link_gbps_assumed": XGMI_LINK_GBPS,
         "exposed_ms": round(with_ms - step_ms, 4),
         "what": f"the step + the gather of O: the block in {stats.get('chunks')} piece(s) (<= {chunks_asked} requested), " + what}
  if pure_ms is not None:
    rep["alone_ms"] = round(pure_ms, 4)
    if expected > 0:
      rep["alone_over_expected"] = round(pure_ms / expected, 3)
      rep["alone_gbps_per_link"] = round(shard_bytes / (pure_ms * 1e-3) / 1e9, 1)
  return rep


def guarded_extra_leg(leg, timeout_s, emit):
  """An OPTIONAL extra figure must never cost the line its contract figure.  ``leg()`` returns the extra object; ``emit(extra)`` prints the (already
  complete) line with it — on rank 0; a no-op elsewhere.  An exception in the leg is recorded in the line.  A leg that does not come back (the gather
  of O has only ever run at world size 1 on hardware: one rank failing leaves the others in a barrier) ends THIS process after ``timeout_s`` with exit
  code 0 — every rank runs the same watchdog —, the line printed without the figure."""
  once = threading.Lock()

  def give_up():
    if once.acquire(blocking=False):
      emit({"error": f"no result within {timeout_s:g} s (watchdog): the line is the sharded step's alone"})
      sys.stdout.flush()
      os._exit(0)

  watchdog = threading.Timer(timeout_s, give_up)
  watchdog.daemon = True
  watchdog.start()
  try:
    extra = leg()
  except Exception as e:  # noqa: BLE001
    extra = {"error": f"{type(e).__name__}: {e}"[:300]}
  if not once.acquire(blocking=False):
    time.sleep(3600)  # (the watchdog got there first: it is printing and about to end the process)
  watchdog.cancel()
  emit(extra)


def main() -> None:
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--warmup", type=int, default=5)
  ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
  ap.add_argument("--gather", action="store_true", help="include an RCCL all_gather of O in the timed step (N > 1)")
  ap.add_argument("--gather-chunks", type=int, default=4, help="--gather: pieces the local block is attended / gathered in (overlap)")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-sdpa", action="store_true", help="skip the SDPA-on-GPU accuracy / speed comparison")
  ap.add_argument("--no-ref-protocol", action="store_true", help="skip the reference bench's own timing protocol (2 warm-ups + 10 iterations)")
  ap.add_argument("--no-gather-extra", action="store_true", help="N > 1: skip the second timed region that adds the all_gather of O")
  ap.add_argument("--gather-extra-timeout", type=float, default=180.0, help="N > 1: seconds the gather leg may take before every rank gives up on it and rank 0 prints the line without it")
  ap.add_argument("--no-live-traffic", action="store_true", help="N = 1: do not measure roofline.traffic with two rocprofv3 PMC passes of this run (quote the committed profile instead)")
  ap.add_argument("--no-steady", action="store_true", help="N = 1: skip the steady-state leg (the step again after >= 150 ms of continuous load, outside the timed region)")
  ap.add_argument("--sweep", action="store_true", help="the reference bench's case table for --sweep-dims in one process (python -m ffpa_attn.bench)")
  ap.add_argument("--sweep-dims", default="320,512,1024")
  ap.add_argument("--stub-backend", default="", help="(tests) run the launch / barrier / reduction plumbing on CPU over this torch.distributed "
                  "backend (gloo) with a sleep for a step: no GPU, no kernel, NOT a benchmark")
  args = ap.parse_args()

  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    # no launcher around us: be our own (the torchrun form keeps working: it sets WORLD_SIZE)
    if not args.stub_backend and torch.cuda.device_count() < args.gpus:
      sys.exit(f"bench.py: --gpus {args.gpus} but this node shows {torch.cuda.device_count()} GPU(s)")
    sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if world != args.gpus:
    sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
  if args.stub_backend:
    return stub_main(args, world, rank)
  if not torch.cuda.is_available():
    sys.exit("bench.py needs a GPU (the HIP kernel has no CPU fallback)")
  if args.sweep:
    return sweep_main(args)
  if args.workload in ("varlen", "varlen_decode"):
    if world != 1:
      sys.exit(f"bench.py: --workload {args.workload} is a single-GPU workload")
    return varlen_main(args) if args.workload == "varlen" else varlen_decode_main(args)
  torch.cuda.set_device(local_rank)
  dev = torch.device("cuda", local_rank)
  dist = None
  if world > 1:
    import torch.distributed as dist

    dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm

  from ffpa_attn_amd import hip, sharding

  hip.load_library()  # fail loudly when the HIP extension is missing
  name = args.workload
  w = WORKLOADS[name]
  B, Hq, Hkv, Nq, Nkv, D = (w[x] for x in ("B", "Hq", "Hkv", "Nq", "Nkv", "D"))
  scale = D ** -0.5
  strong = name == "cfg5"
  sharded = world > 1 or strong
  grp = Hq // Hkv
  mask = make_mask(w, torch.bfloat16, dev)

  if sharded:
    # born-sharded: this rank's contiguous block of (batch, kv-head) units of the global problem, per-unit seeds
    global_B = B if strong else B * world
    n_units = global_B * Hkv
    u0, u1 = sharding.partition_units(n_units, world, rank)
    q, k, v = sharding.synthetic_unit_block(u0, u1, grp, Nq, Nkv, D, device=dev, seed=0)
    local_units = u1 - u0
    flops_local = valid_pairs_flops(w, 1) // Hkv * local_units  # FLOPs are uniform over the units
    flops_global = valid_pairs_flops(w, global_B)
    want_gather_extra = world > 1 and not args.gather and not args.no_gather_extra
    gathered = torch.empty((n_units, grp, Nq, D), dtype=torch.bfloat16, device=dev) if (world > 1 and (args.gather or want_gather_extra)) else None
    api_kw = dict(is_causal=w["causal"], dropout_p=w["dropout"])
    gather_stats = {}
    if mask is not None:
      api_kw["attn_mask"] = mask

    def step_kernel():
      if w["via"] == "op_offset0":  # the op's structured top-left causal mask (the public is_causal is tail-aligned and needs Nkv >= Nq)
        return hip.ffpa_attn_forward_hip(q, k, v, None, causal=True, softmax_scale=scale, causal_offset=0)[0]
      return sharding.attend_units(q, k, v, **api_kw)

    def step_gather():
      if w["via"] == "op_offset0":
        o = hip.ffpa_attn_forward_hip(q, k, v, None, causal=True, softmax_scale=scale, causal_offset=0)[0]
        return sharding.gather_units(o, n_units, out=gathered)
      # the block in pieces, each piece all-gathered (RCCL's stream) while the next one computes
      return sharding.attend_and_gather_units(q, k, v, n_units, chunks=args.gather_chunks, out=gathered, stats=gather_stats, **api_kw)

    step = step_gather if (args.gather and world > 1) else step_kernel
  else:
    global_B = B
    torch.manual_seed(0)
    q = torch.randn(B, Hq, Nq, D, dtype=torch.bfloat16, device=dev)
    k = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device=dev)
    v = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device=dev)
    flops_local = flops_global = valid_pairs_flops(w, B)
    gathered = None
    want_gather_extra = False
    if w["via"] == "op_offset0":
      def step():
        return hip.ffpa_attn_forward_hip(q, k, v, None, causal=True, softmax_scale=scale, causal_offset=0)[0]
    else:
      def step():
        return ffpa_attn_func(q, k, v, attn_mask=mask, dropout_p=w["dropout"], is_causal=w["causal"], enable_gqa=Hq != Hkv)

      if w["bound"] == "hbm" and w["dropout"] == 0.0:
        # the same step through the product's graph-replayed form (ffpa_attn_amd.DecodeStep: the call captured once, one hipGraphLaunch per step): timed next to
        # the contract figure (`decode_step` in the line).  Measured (profiles/r06_decode_step.txt): it takes 19 us of HOST time off a call (33 -> 14) and 15 us off
        # a caller that synchronises per token; a back-to-back launch loop like the timed region is GPU-bound either way (a replay costs the GPU ~ 3 us of launch).
        from ffpa_attn_amd import DecodeStep

        _decode_step = DecodeStep(is_causal=w["causal"], enable_gqa=Hq != Hkv)

        def step_graph():
          return _decode_step(q, k, v, mask)

  def timed(fn):
    """W untimed warm-ups, then EXACTLY K steps between barrier + synchronize on both sides; elapsed = max over ranks."""
    for _ in range(args.warmup):
      fn()
    torch.cuda.synchronize()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    if dist is not None:
      dist.barrier()
    torch.cuda.synchronize()
    telemetry.start()
    t0 = time.perf_counter()
    for i in range(args.steps):
      starts[i].record()  # the kernel is launched on torch's current stream; so are these events
      fn()
      ends[i].record()
    torch.cuda.synchronize()
    local = time.perf_counter() - t0
    telemetry.stop()
    if dist is not None:
      dist.barrier()
    total = time.perf_counter() - t0
    per_rank = None
    if dist is not None:
      t = torch.tensor([total], dtype=torch.float64, device=dev)
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
      total = float(t.item())
      mine = torch.tensor([flops_local * args.steps / local / 1e12], dtype=torch.float64, device=dev)
      every = [torch.zeros_like(mine) for _ in range(world)]
      dist.all_gather(every, mine)
      per_rank = [round(float(x.item()), 2) for x in every]
    return total, per_rank, sorted(a.elapsed_time(b) for a, b in zip(starts, ends))

  telemetry = DeviceTelemetry(local_rank)
  elapsed, per_rank_tflops, kernel_ms = timed(step)
  device = telemetry.summary()  # (of the first timed region: the bench line's own)
  kernel_ms_avg = sum(kernel_ms) / len(kernel_ms)
  value = flops_global * args.steps / elapsed / 1e12
  steady = None
  if world == 1 and not args.no_steady and not args.stub_backend:
    # Context for `value`, never `value` itself: a GPU coming out of idle runs its first ~ 10 - 25 ms of work slower than it does under
    # continuous load (profiles/r04_clock_ramp.txt: the first 25 launches of a 0.44 ms kernel take 0.53 ms at a REPORTED 2.4 GHz, a 3.2 ms
    # kernel's first five 3.7 ms), and socket power takes ~ 200 ms to reach its cap, pulling the clock from 2.4 to ~ 2.0 GHz.  W = 5 + K = 20
    # steps of a short kernel live inside that transient.  This leg repeats the step back to back until >= 150 ms of load have passed and
    # times the launches after that with one event pair.
    est_ms = max(kernel_ms_avg, 1e-3)
    n_pre, n_timed = max(20, int(150.0 / est_ms)), max(20, min(400, int(60.0 / est_ms)))
    for _ in range(n_pre):
      step()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    telemetry.start()
    ev0.record()
    for _ in range(n_timed):
      step()
    ev1.record()
    torch.cuda.synchronize()
    telemetry.stop()
    tel2 = telemetry.summary()
    ss_ms = ev0.elapsed_time(ev1) / n_timed
    steady = {"ms_per_step": round(ss_ms, 4), "tflops": round(flops_local / ss_ms / 1e9, 2), "launches": n_timed, "after_launches": n_pre,
              "sclk_mhz_avg": tel2.get("sclk_mhz_avg"), "power_w_avg": tel2.get("power_w_avg"),
              "what": "the same step back to back after >= 150 ms of continuous load, one HIP event pair around the launches; outside the timed region"}
    if w["bound"] == "hbm":
      steady["gbps"] = round(algorithmic_bytes(w, global_B) / (ss_ms * 1e-3) / 1e9, 1)
  other_launches = None
  if world == 1 and not sharded and w.get("one_range_leg") and not args.stub_backend:
    # Context, never `value`: the same call with ONE KV range per row tile (num_splits = 1: the launch before the per-row-tile ranges), interleaved with the plan's launch
    def one_range():
      return hip.forward(q, k, v, None, w["causal"], scale, num_splits=1, return_lse=False)[0]

    def plan_launch():
      return hip.forward(q, k, v, None, w["causal"], scale, return_lse=False)[0]

    legs = {"one_kv_range": [], "plan": []}
    for _ in range(7):
      for label, fn in (("one_kv_range", one_range), ("plan", plan_launch)):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(10):
          fn()
        ev1.record()
        torch.cuda.synchronize()
        legs[label].append(ev0.elapsed_time(ev1) / 10)
    p1 = {}
    hip.forward(q, k, v, None, w["causal"], scale, num_splits=1, return_lse=False, plan_out=p1)
    med = {k_: sorted(v_)[3] for k_, v_ in legs.items()}
    other_launches = {"what": "interleaved: 7 x 10 launches per arm through hip.forward, median; outside the timed region",
                      "one_kv_range": {"ms": round(med["one_kv_range"], 4), "tflops": round(flops_local / med["one_kv_range"] / 1e9, 1), "kernel": p1.get("kernel")},
                      "plan": {"ms": round(med["plan"], 4), "tflops": round(flops_local / med["plan"] / 1e9, 1)},
                      "plan_over_one_range": round(med["one_kv_range"] / med["plan"], 3)}
  decode_step = None
  if world == 1 and not sharded and "step_graph" in locals():
    try:
      for _ in range(args.warmup):
        step_graph()
      torch.cuda.synchronize()
      t1 = time.perf_counter()
      for _ in range(args.steps):
        step_graph()
      torch.cuda.synchronize()
      ds_ms = (time.perf_counter() - t1) * 1e3 / args.steps
      sync_ms = {}
      for label, fn in (("DecodeStep", step_graph), ("ffpa_attn_func", step)):  # a caller that synchronises after every step: host + GPU, nothing overlaps
        t1 = time.perf_counter()
        for _ in range(100):
          fn()
          torch.cuda.synchronize()
        sync_ms[label] = round((time.perf_counter() - t1) * 1e3 / 100, 4)
      decode_step = {"ms_per_step": round(ds_ms, 4), "gbps": round(algorithmic_bytes(w, global_B) / (ds_ms * 1e-3) / 1e9, 1), "ms_per_step_synchronising_caller": sync_ms,
                     "what": f"ffpa_attn_amd.DecodeStep (the call captured into a HIP graph once, replayed per step), {args.warmup} warm-ups + {args.steps} steps, perf_counter around a synchronize; outside the timed region"}
    except Exception as e:  # noqa: BLE001 — informative only
      decode_step = {"error": f"{type(e).__name__}: {e}"[:300]}
  graph_replay = None
  if world == 1 and w["bound"] == "hbm" and not args.no_steady and not args.stub_backend:
    # Context, never `value`: the decode step is launch-bound on the host (two kernels of ~ 85 + 5 us behind ~ 30 us of Python per call).  A serving loop captures its
    # layers into a HIP graph; this leg captures the step once — and 32 of them, a token's worth of layers — into torch.cuda.CUDAGraph and replays back to back: what the
    # kernels alone sustain when nothing is launched from Python (the C-ABI allocates nothing, never synchronises and launches on the caller's stream, so it captures as is).
    try:
      graph_replay = {"what": "the step captured into a HIP graph (torch.cuda.CUDAGraph) and replayed back to back after >= 150 ms of load; outside the timed region"}
      side = torch.cuda.Stream()
      side.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(side):
        for _ in range(3):
          step()
      torch.cuda.current_stream().wait_stream(side)
      for per_graph in (1, 32):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
          for _ in range(per_graph):
            keep = step()  # noqa: F841 — (the outputs live in the graph's pool)
        est_ms = max(kernel_ms_avg, 1e-3) * per_graph
        n_pre, n_timed = max(3, int(150.0 / est_ms)), max(5, min(400, int(60.0 / est_ms)))
        for _ in range(n_pre):
          g.replay()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(n_timed):
          g.replay()
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / (n_timed * per_graph)
        graph_replay[f"steps_per_graph_{per_graph}"] = {"ms_per_step": round(ms, 4), "gbps": round(algorithmic_bytes(w, global_B) / (ms * 1e-3) / 1e9, 1),
                                                       "frac_of_hbm_peak": round(algorithmic_bytes(w, global_B) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "replays": n_timed}
        del g
    except Exception as e:  # noqa: BLE001 — informative only
      graph_replay = {"error": f"{type(e).__name__}: {e}"[:300]}
  # (the whole local problem, not a slice of it: since round 5 the launch plan depends on how many workgroups a launch has — the wide-row tile of D = 320)
  plan = planned_kernel(w, q, k, v, mask, scale) if rank == 0 else {}

  if rank == 0:
    build = build_identity()
    traffic, traffic_src, traffic_stale = measured_traffic(name, build.get("lib_sha16")) if world == 1 else (None, None, False)
    traffic_live_note = None
    # (never under a profiler of somebody else's: tools/gpu_round.sh runs this script under rocprofv3 itself — a nested profiler would fight it for the counters)
    under_profiler = any(k_.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k_ in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "").lower()
    if world == 1 and not sharded and not args.no_live_traffic and not args.stub_backend and not under_profiler:
      torch.cuda.synchronize()
      live, why = live_traffic(name)
      if live is not None:
        traffic, traffic_src, traffic_stale = live, why, False
      else:
        traffic_live_note = why
    if w["bound"] == "hbm":
      bytes_launch = algorithmic_bytes(w, global_B)
      achieved = bytes_launch / (kernel_ms_avg * 1e-3) / 1e9
      roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
              "traffic": traffic, "traffic_source": traffic_src, "kernel": plan.get("kernel"),
              "kernel_ms_avg": round(kernel_ms_avg, 4), "kernel_ms_median": round(kernel_ms[len(kernel_ms) // 2], 4), "bytes_per_launch": bytes_launch}
    else:
      achieved = flops_local / (kernel_ms_avg * 1e-3) / 1e12
      roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
              "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
              "kernel": plan.get("kernel"), "kernel_ms_avg": round(kernel_ms_avg, 4),
              "kernel_ms_median": round(kernel_ms[len(kernel_ms) // 2], 4), "flops_per_launch": flops_local,
              "algorithmic_bytes_per_launch": algorithmic_bytes(w, max(1, (u1 - u0) // Hkv) if sharded else B)}
      # the same figure against what THIS device says it is, and per clock at the clock it actually ran: two boxes of a pool differ by
      # their clock under load, not by the kernel (4096 = 4 SIMDs x 1024 bf16 MFMA FLOP per CU and clock)
      if device.get("peak_tflops_from_device"):
        roof["frac_of_device_peak"] = round(achieved / device["peak_tflops_from_device"], 4)
      if device.get("sclk_mhz_avg"):
        roof["flop_per_clk_per_cu"] = round(achieved * 1e12 / (device["cus"] * device["sclk_mhz_avg"] * 1e6), 1)
        roof["frac_of_mfma_rate_at_measured_clock"] = round(roof["flop_per_clk_per_cu"] / 4096.0, 4)
    roof["traffic_stale"] = bool(traffic_stale)
    if traffic_live_note is not None:
      roof["traffic_live_failed"] = traffic_live_note  # (the figure above is then the committed profile's, if its library is this one)
    shape = f"B={global_B} Hq={Hq} Hkv={Hkv} Nq={Nq} Nkv={Nkv} D={D}"
    line = {
      "metric": metric_name(name, w),
      "value": round(value, 2),
      "unit": "TFLOPS",
      "n_gpus": world,
      "steps": args.steps,
      "warmup": args.warmup,
      "ms_per_step": round(elapsed / args.steps * 1e3, 4),
      "higher_is_better": True,
      "scaling": "strong" if strong else "weak",
      "vs_baseline": None,  # BASELINE.md holds no published MI355X number for this metric
      "dtype": "bf16",
      "data": "synthetic",
      "config": {
        "workload": f"{name}: {shape} bf16 attention forward" + (f" ({w['note']})" if w["note"] else ""),
        "global_batch": global_B,
        "seq_len": Nq,
        "parallelism": (f"(batch,kv-head) units born sharded x{world}: {Hkv * global_B} units, {Hkv * global_B / world:g} per rank, no data-path collective"
                        if sharded else "single GPU") + (f" + gather of O on every rank in <= {args.gather_chunks} pieces overlapped with compute" if (sharded and world > 1 and args.gather) else ""),
        "flops_model": "4*B*Hq*D*valid_pairs",
        "step": "hip.ffpa_attn_forward_hip(causal=True, causal_offset=0)" if w["via"] == "op_offset0" else
                ("sharding.attend_units -> ffpa_attn_func" if sharded else "ffpa_attn_func"),
      },
      "roofline": roof,
      "device": device,
      "steady_state": steady,
      **({"decode_step": decode_step} if decode_step is not None else {}),
      **({"other_launches": other_launches} if other_launches is not None else {}),
      **({"graph_replay": graph_replay} if graph_replay is not None else {}),
      "build": build,
      "plan": {k_: plan.get(k_) for k_ in ("variant", "block_rows", "block_keys", "splits")},
    }
    if per_rank_tflops is not None:
      line["per_rank_tflops"] = per_rank_tflops
      line["rccl_world_size"] = world
      line["frac_of_mfma_peak_aggregate"] = round(value / (MFMA_BF16_PEAK_TFLOPS * world), 4)
      line["timed_step_includes_gather"] = bool(args.gather)
    if world == 1 and not sharded and not args.no_ref_protocol:
      # the reference bench's protocol: 2 warm-ups, 10 iterations, perf_counter around a synchronize (cli/_runner_fwd.py:84-103)
      for _ in range(2):
        step()
      torch.cuda.synchronize()
      t1 = time.perf_counter()
      for _ in range(10):
        step()
      torch.cuda.synchronize()
      ref_ms = (time.perf_counter() - t1) * 1e3 / 10
      line["ref_protocol"] = {"ms": round(ref_ms, 4), "tflops": round(flops_global / ref_ms / 1e9, 2), "warmup": 2, "iters": 10,
                              "timer": "perf_counter + cuda.synchronize (cli/_runner_fwd.py:84-103)"}
    if world == 1 and not sharded and not args.no_sdpa and w["dropout"] == 0.0:
      try:
        acc = accuracy(w, q, k, v, mask, scale)
        line.update(acc)
        line["speedup_vs_sdpa_gpu"] = round(acc["sdpa_gpu_ms"] / (elapsed / args.steps * 1e3), 3)
      except Exception as e:  # noqa: BLE001 - the comparison is informative, the metric above is not
        line["sdpa_error"] = str(e)[:200]
    elif w["dropout"] > 0.0:
      line["accuracy_note"] = ("dropout: PyTorch-ROCm's fused SDPA draws another Philox stream, so outputs are not comparable element-wise "
                               "(the reference skips its dropout-parity test on ROCm, tests/test_ffpa_fwd.py:339-342); the mask is pinned "
                               "bit-for-bit against the oracle in tests/test_fwd_gpu.py")
    if world == 1 and not args.no_cpu_baseline:
      line["cpu_baseline"] = cpu_baseline(w)
  else:
    line = None

  def emit(extra=None):
    if line is not None:
      if extra is not None:
        line["with_gather"] = extra
      print(json.dumps(line), flush=True)

  if want_gather_extra:
    # The same K steps once more with the gather of O inside the step: both figures in one line.  It runs LAST, with the line already complete.
    def gather_leg():
      g_elapsed, g_per_rank, _ = timed(step_gather)
      # the gather alone (nothing to hide behind): one all_gather_into_tensor of the finished block, the same K steps
      o_done = step_kernel()
      p_elapsed, _, _ = timed(lambda: sharding.gather_units(o_done, n_units, out=gathered))
      shard_bytes = o_done.numel() * o_done.element_size()
      rep = gather_report(gather_stats, shard_bytes, world, elapsed / args.steps * 1e3, g_elapsed / args.steps * 1e3, p_elapsed / args.steps * 1e3, args.gather_chunks)
      return {"value": round(flops_global * args.steps / g_elapsed / 1e12, 2), "unit": "TFLOPS", "ms_per_step": round(g_elapsed / args.steps * 1e3, 4),
              "per_rank_tflops": g_per_rank, **rep}

    guarded_extra_leg(gather_leg, args.gather_extra_timeout, emit)
  else:
    emit()

  if dist is not None:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
