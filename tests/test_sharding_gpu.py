"""The multi-GPU path with the real kernel over RCCL (`pytest -m gpu`).

`ffpa_attn_amd.sharding` born-sharded flow — per-unit seeded blocks -> HIP kernel on the local block -> one
all_gather_into_tensor of O — as `bench.py --workload cfg5` runs it.  World size 1 always runs (one MI355X: the
"nccl" backend is RCCL, the collective degenerates to a copy but goes through the same calls); world size 2 runs
when the box has two GPUs (the driver's 8-GPU node), and must reproduce the one-rank result bit for bit — units are
independent, so the partition cannot change a single output element.
"""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_UNITS, GROUP, NQ, NKV, D = 8, 4, 640, 1024, 512  # 8 (batch, kv-head) units of 4 query heads each


def _free_port():
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


def _worker(rank, world, port, ret, transport=""):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", FFPA_SHARDING_TRANSPORT=transport)
  torch.cuda.set_device(rank)
  dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
  from ffpa_attn_amd import hip
  from ffpa_attn_amd import sharding as sh

  hip.load_library()  # no fallback: the extension must be there
  s, e = sh.local_units(N_UNITS)
  q, k, v = sh.synthetic_unit_block(s, e, GROUP, NQ, NKV, D, device=f"cuda:{rank}", seed=11)
  o_local = sh.attend_units(q, k, v)
  full = sh.gather_units(o_local, N_UNITS)
  st = {}
  overlapped = sh.attend_and_gather_units(q, k, v, N_UNITS, chunks=2, stats=st)  # pieces gathered on RCCL's stream while the next computes
  assert st["transport"] == (transport or ("p2p" if world > 1 else "local")) or (world > 1 and not transport and st["transport"] == "all_gather"), st
  whole = sh.attend_and_gather_units(q, k, v, N_UNITS, chunks=1)
  torch.cuda.synchronize()
  assert torch.equal(whole, full)
  # a piece of this small problem would under-fill the chip (its launch would split the KV axis: fp32 partials + LSE merge, the same values
  # to rounding only) — attend_and_gather_units lowers the piece count instead, so the gathered tensor does not depend on `chunks`
  assert torch.equal(overlapped, full)
  # the overlapped form writes every piece straight into its final slice (its own: one copy; the others': point-to-point receives) — no staging
  # buffer of world x piece next to the result: the peak above the buffers the caller holds is this rank's own pieces + the call's scratch
  mine = torch.empty_like(full)
  torch.cuda.synchronize()
  torch.cuda.reset_peak_memory_stats()
  base = torch.cuda.memory_allocated()
  sh.attend_and_gather_units(q, k, v, N_UNITS, chunks=2, out=mine)
  torch.cuda.synchronize()
  peak = torch.cuda.max_memory_allocated() - base
  own = o_local.numel() * o_local.element_size()
  assert torch.equal(mine, full)
  pl = hip.launch_plan(q.size(0), GROUP, 1, NQ, NKV, D, device=q.device)  # (a launch this small splits the KV axis: fp32 partials + LSE as scratch)
  scratch = pl["splits"] * q.size(0) * GROUP * NQ * (D + 1) * 4 if pl["splits"] > 1 else 0
  if st["transport"] != "all_gather":  # (the fallback transport IS the world x piece temporary)
    assert peak <= own + scratch + (1 << 20), (peak, own, scratch)  # a world x piece staging buffer (2 x own at two ranks) would not fit the slack
  ref = torch.nn.functional.scaled_dot_product_attention(q, k, v, enable_gqa=True)
  err = (o_local.float() - ref.float()).abs().max().item()
  ret[rank] = (full.cpu(), err, (s, e), st["transport"])
  dist.barrier()
  dist.destroy_process_group()


RANKS_TIMEOUT_S = 420  # a first contact with N > 1 ranks must end in a verdict, not in a hung collective


def _run(world, transport=""):
  """`world` ranks of _worker under a deadline of their own: ranks that do not come back (a collective one of them never entered) are killed and the
  test FAILS with words — pytest's own timeout would take the whole session down with it."""
  import time

  ret = mp.Manager().dict()
  ctx = mp.spawn(_worker, args=(world, _free_port(), ret, transport), nprocs=world, join=False)
  deadline = time.time() + RANKS_TIMEOUT_S
  while not ctx.join(timeout=5):
    if time.time() > deadline:
      for pr in ctx.processes:
        if pr.is_alive():
          pr.kill()
      pytest.fail(f"{world} ranks over RCCL did not finish within {RANKS_TIMEOUT_S} s (transport {transport or 'auto'}): killed")
  return ret


def test_one_rank_rccl_born_sharded_flow_with_the_hip_kernel():
  ret = _run(1)
  full, err, span, transport = ret[0]
  assert transport == "local" and span == (0, N_UNITS) and full.shape == (N_UNITS, GROUP, NQ, D)
  assert err <= 1e-2  # the north star's max-abs bound vs SDPA on the same inputs


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (runs on the driver's multi-GPU node)")
def test_two_ranks_over_rccl_reproduce_the_one_rank_result_bit_for_bit():
  one = _run(1)[0][0]
  ret = _run(2)
  assert ret[0][2] == (0, N_UNITS // 2) and ret[1][2] == (N_UNITS // 2, N_UNITS)
  assert torch.equal(ret[0][0], one) and torch.equal(ret[1][0], one)
  assert max(ret[0][1], ret[1][1]) <= 1e-2
  assert ret[0][3] == ret[1][3] and ret[0][3] in ("p2p", "all_gather")  # (the ranks agreed, whatever the node offers)
  # the fallback transport pinned (what the probe selects on a node without RCCL point-to-point): the same bits
  if ret[0][3] == "p2p":
    alt = _run(2, "all_gather")
    assert alt[0][3] == "all_gather" and torch.equal(alt[0][0], one) and torch.equal(alt[1][0], one)


# (`pytest -m gpu -k two_ranks` on a 2-GPU box runs exactly the N > 1 tests)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (runs on the driver's multi-GPU node)")
def test_two_ranks_bench_launches_itself_over_rccl():
  """`python bench.py --gpus 2 --workload cfg5` — the command the scaling curve is taken with — as a subprocess: one JSON line, RCCL world
  size 2, one TFLOPS figure per rank, the kernel-only figure and the figure with the all_gather of O inside the step.  (No number is
  asserted: the plumbing must not be what fails when the curve is taken.)"""
  import json
  import subprocess
  import sys

  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
  for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k_, None)
  out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "cfg5", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=RANKS_TIMEOUT_S + 180, env=env)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
  assert len(lines) == 1, out.stdout[-2000:]
  line = json.loads(lines[0])
  assert line["n_gpus"] == 2 and line["rccl_world_size"] == 2 and line["scaling"] == "strong"
  assert len(line["per_rank_tflops"]) == 2 and all(x > 0 for x in line["per_rank_tflops"])
  assert line["with_gather"]["value"] > 0 and line["value"] > 0 and line["timed_step_includes_gather"] is False
  g = line["with_gather"]  # the self-judging figures of the first multi-GPU run: transport, agreed pieces, measured vs one shard per xGMI link
  assert g["transport"] in ("p2p", "all_gather") and g["chunks"] >= 1 and g["alone_ms"] > 0 and g["expected_ms_one_shard_per_link"] > 0 and g["shard_bytes"] == 128 * 8192 * 512 * 2, g
  assert line["config"]["global_batch"] == 8 and "256 units, 128 per rank" in line["config"]["parallelism"]
