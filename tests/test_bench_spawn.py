"""bench.py's own launcher (CPU, gloo): `python bench.py --gpus N` without torchrun must start N ranks itself, rank 0 must print ONE
JSON line, the reported time must be the slowest rank's; under a launcher (WORLD_SIZE set) it must not spawn again."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env=None):
  e = dict(os.environ)
  for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
    e.pop(k, None)
  e.update(env or {})
  return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=240, env=e)


def test_bench_spawns_its_own_ranks_when_no_launcher_did():
  r = _run(["--gpus", "2", "--stub-backend", "gloo", "--steps", "4", "--warmup", "1"])
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
  assert len(lines) == 1, r.stdout  # rank 0 only
  d = json.loads(lines[0])
  assert d["n_gpus"] == 2 and d["steps"] == 4 and len(d["per_rank_s"]) == 2
  # rank 1 sleeps twice as long per step: the step time is the slowest rank's (max over ranks), not rank 0's
  assert d["per_rank_s"][1] > d["per_rank_s"][0] and d["ms_per_step"] * 4e-3 >= d["per_rank_s"][1] * 0.99


def test_bench_plumbing_at_the_scaling_curves_world_sizes():
  """The driver takes the scaling curve at N = 1, 2, 4, 8: the launch / barrier / max-over-ranks plumbing at 4 and 8 ranks (gloo stand-in; the
  data path needs GPUs and is covered at world size 2 in tests/test_sharding.py and on the GPU box)."""
  for n in (4, 8):
    r = _run(["--gpus", str(n), "--stub-backend", "gloo", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and len(d["per_rank_s"]) == n and d["ms_per_step"] * 2e-3 >= max(d["per_rank_s"]) * 0.99
    # the gather leg's report (round 6): which transport the ranks agreed on, the agreed piece count, and the self-judging figures
    g = d["with_gather"]
    assert g["transport"] == "p2p" and g["chunks"] == 2 and g["chunks_requested"] == 4 and g["equal_to_plain_gather"] is True, g
    assert g["received_bytes_per_rank"] == g["shard_bytes"] * (n - 1) and g["expected_ms_one_shard_per_link"] > 0 and g["alone_ms"] > 0 and "alone_over_expected" in g, g
  # the fallback transport, pinned through the environment (what the probe selects on a node without RCCL point-to-point)
  r = _run(["--gpus", "4", "--stub-backend", "gloo", "--steps", "2", "--warmup", "1"], env={"FFPA_SHARDING_TRANSPORT": "all_gather"})
  assert r.returncode == 0, r.stderr[-2000:]
  g = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["with_gather"]
  assert g["transport"] == "all_gather" and g["equal_to_plain_gather"] is True and "all_gather_into_tensor" in g["what"], g


def test_bench_under_a_launcher_does_not_spawn_and_checks_the_world_size():
  port = "29631"
  r = _run(["--gpus", "1", "--stub-backend", "gloo", "--steps", "2", "--warmup", "0"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port})
  assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1, r.stderr[-1000:]
  r = _run(["--gpus", "2", "--stub-backend", "gloo"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port})
  assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_a_failing_rank_fails_the_launcher_promptly(tmp_path):
  """A rank that dies must take the launch down at once (the other ranks would wait for it in the rendezvous for minutes): rank 1 of a
  stub launch is made to crash through an unusable backend name for that rank only."""
  import time

  # no GPU here: the real (non-stub) path refuses before spawning when the node has fewer GPUs than ranks
  r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], env={"CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""})
  assert r.returncode != 0 and ("GPU(s)" in r.stderr or "needs a GPU" in r.stderr), r.stderr[-500:]
  # the polling loop itself, driven directly: one child exits 3 at once, the other would sleep for a minute
  sys.path.insert(0, ROOT)
  import bench

  script = tmp_path / "rank.py"
  script.write_text("import os, sys, time\nsys.exit(3) if os.environ['RANK'] == '1' else time.sleep(60)\n")
  real = bench.__file__
  bench.__file__ = str(script)
  try:
    t0 = time.time()
    rc = bench.spawn_ranks(2, [])
  finally:
    bench.__file__ = real
  assert rc == 3 and time.time() - t0 < 20


def _leg(code: str):
  e = dict(os.environ, PYTHONPATH=ROOT)
  return subprocess.run([sys.executable, "-c", "import json, sys, time\nimport bench\n" + code], capture_output=True, text=True, timeout=240, env=e, cwd=ROOT)


def test_an_optional_leg_cannot_cost_the_line():
  """N > 1: the gather of O is timed after the contract figure and under a watchdog (bench.guarded_extra_leg) — it returns, raises or hangs; the
  line is printed exactly once and the process ends with exit code 0 in all three cases."""
  emit = "emit = lambda extra: print(json.dumps({'value': 1.0, 'with_gather': extra}), flush=True)\n"
  r = _leg(emit + "bench.guarded_extra_leg(lambda: {'value': 2.0}, 30.0, emit)")
  assert r.returncode == 0 and [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")] == [{"value": 1.0, "with_gather": {"value": 2.0}}], r.stdout + r.stderr
  r = _leg(emit + "bench.guarded_extra_leg(lambda: 1 / 0, 30.0, emit)")
  (d,) = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
  assert r.returncode == 0 and d["value"] == 1.0 and d["with_gather"]["error"].startswith("ZeroDivisionError"), r.stdout + r.stderr
  r = _leg(emit + "bench.guarded_extra_leg(lambda: time.sleep(600), 0.5, emit)\nprint('{\"never\": 1}')")
  (d,) = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
  assert r.returncode == 0 and d["value"] == 1.0 and "watchdog" in d["with_gather"]["error"], r.stdout + r.stderr


def test_gather_report_prices_a_shard_per_link():
  sys.path.insert(0, ROOT)
  import bench

  rep = bench.gather_report({"transport": "p2p", "chunks": 4, "world": 8}, 256 << 20, 8, 3.2, 3.5, 2.1, 4)
  assert abs(rep["expected_ms_one_shard_per_link"] - 1.7545) < 1e-3 and rep["received_bytes_per_rank"] == 7 * (256 << 20)  # 256 MiB / 153 GB/s = 1.75 ms
  assert abs(rep["exposed_ms"] - 0.3) < 1e-9 and rep["alone_ms"] == 2.1 and abs(rep["alone_over_expected"] - 1.197) < 1e-3 and "point-to-point" in rep["what"]
  assert "unavailable" in bench.gather_report({"transport": "all_gather", "chunks": 1}, 1 << 20, 2, 1.0, 1.2, None, 4)["what"]
