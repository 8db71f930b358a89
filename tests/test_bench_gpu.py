"""bench.py as the driver runs it, on one GPU: one JSON line with the contract's keys, the roofline and CPU-baseline objects, what the device
said about itself, and the steady-state leg.  No throughput is asserted — only that the line is whole and self-consistent."""

import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(*flags):
  env = dict(os.environ)
  for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k_, None)
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=600, env=env)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
  assert len(lines) == 1, out.stdout[-2000:]
  return json.loads(lines[0])


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_bench_line_of_a_short_workload_is_whole():
  line = _bench("--gpus", "1", "--workload", "cross", "--steps", "4", "--warmup", "2", "--no-sdpa")
  for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "device", "steady_state", "build", "plan"):
    assert key in line, key
  assert line["n_gpus"] == 1 and line["steps"] == 4 and line["warmup"] == 2 and line["dtype"] == "bf16" and line["vs_baseline"] is None
  assert "model" not in line["config"] and line["config"]["workload"].startswith("cross")
  roof = line["roofline"]
  assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 2500.0 and "traffic" in roof
  # round 6: the HBM traffic of the dominant kernel is MEASURED in the run (two rocprofv3 PMC passes of the same command behind the timed region) — at least the
  # compulsory bytes, and nowhere near the operand stream the kernel re-reads from L2 (a counter in the wrong unit would be off by 1024 or more)
  if "traffic_live_failed" not in roof:
    assert roof["traffic_source"].startswith("measured in this run") and roof["traffic_stale"] is False
    assert 0.8 * roof["algorithmic_bytes_per_launch"] <= roof["traffic"] <= 20 * roof["algorithmic_bytes_per_launch"], roof
  else:
    assert isinstance(roof["traffic_live_failed"], str) and roof["traffic_live_failed"]
  assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and 0.0 < roof["frac"] < 1.0
  assert roof["kernel"].startswith("ffpa_fwd_m16_kernel<bf16, 512")
  # the step cannot be shorter than its kernel, and the value follows from the step time
  assert roof["kernel_ms_avg"] <= line["ms_per_step"] * 1.02
  assert abs(line["value"] - roof["flops_per_launch"] / line["ms_per_step"] / 1e9) / line["value"] < 0.01
  cpu = line["cpu_baseline"]
  assert cpu["kind"] in ("reference", "port") and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]
  assert line["device"]["cus"] == torch.cuda.get_device_properties(0).multi_processor_count
  steady = line["steady_state"]
  assert steady["launches"] >= 20 and steady["after_launches"] >= 20 and steady["ms_per_step"] > 0
  assert abs(steady["tflops"] - roof["flops_per_launch"] / steady["ms_per_step"] / 1e9) / steady["tflops"] < 0.01
  assert line["build"]["lib"].endswith("libffpa_attn_hip.so") and len(line["build"]["lib_sha16"]) == 16


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_decode_line_is_priced_against_hbm_and_carries_the_graph_replay_leg():
  """The decode workload's roofline is the HBM one; its step is launch-bound from Python, so the line also says what the same step does when it is
  captured into a HIP graph and replayed (one step per graph, and 32 — a token's layers)."""
  line = _bench("--gpus", "1", "--workload", "decode", "--steps", "10", "--warmup", "3", "--no-sdpa", "--no-cpu-baseline", "--no-live-traffic")
  roof = line["roofline"]
  assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0 and 0.0 < roof["frac"] < 1.0
  assert roof["kernel"].startswith("ffpa_fwd_split_d_kernel<bf16, 512")
  g = line["graph_replay"]
  assert "error" not in g, g
  for key in ("steps_per_graph_1", "steps_per_graph_32"):
    leg = g[key]
    assert leg["ms_per_step"] > 0 and abs(leg["gbps"] - roof["bytes_per_launch"] / leg["ms_per_step"] / 1e6) / leg["gbps"] < 0.01 and 0.0 < leg["frac_of_hbm_peak"] < 1.0
  # no host work per step: a replayed step cannot be slower than the step launched from Python by more than timer noise
  assert g["steps_per_graph_32"]["ms_per_step"] <= line["ms_per_step"] * 1.10


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_packed_decode_line_is_priced_against_hbm_and_names_its_launch():
  """`--workload varlen_decode`: a packed decode batch (one token per sequence, ragged KV lengths, GQA) — the HBM roofline of one step, the launch the library
  picked (heads packed into tile rows, KV ranges split and merged, the non-temporal fetch) and the same call without each of them beside it."""
  line = _bench("--gpus", "1", "--workload", "varlen_decode", "--steps", "4", "--warmup", "2", "--no-live-traffic")
  for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
              "device", "steady_state", "build", "plan", "other_launches", "per_sequence_loop"):
    assert key in line, key
  roof = line["roofline"]
  assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0 and 0.05 < roof["frac"] < 1.0
  assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["kernel_ms_avg"] * 1e-3) / 1e9) <= 0.01 * roof["achieved"]
  assert "packed into rows" in roof["kernel"] and line["plan"]["workgroups"] == 32 * 8 * line["plan"]["splits"]
  legs = line["other_launches"]
  assert legs["one_kv_range"]["splits"] == 1 and legs["one_workgroup_per_query_head"]["workgroups"] % (32 * 32) == 0 and ", NT>" not in legs["no_nt_hint"]["kernel"]
  assert line["per_sequence_loop"]["max_abs_diff_vs_packed"] < 4e-3 and line["max_abs_err_vs_sdpa"] < 1e-2
