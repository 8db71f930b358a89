"""bench.py — the reference's headline benchmark on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): attention forward TFLOPS + max-abs-err vs SDPA, bf16 B=1 H=32 N=8192 D=512.
One "step" = one pass of the hot path (ffpa_attn_func -> ffpa_attn::_fwd_hip -> C-ABI -> HIP kernel)
over one synthetic batch already resident in HBM.  FLOPs = 4*B*Hq*D*valid_pairs, the reference's own
model (src/ffpa_attn/cli/_flops.py:37-53); inputs are seed-0 randn, q then k then v
(src/ffpa_attn/cli/_runner_fwd.py:344-347).

N > 1: the path is embarrassingly parallel over (batch, kv-head) (ffpa_attn_amd/sharding.py), so every
rank owns one BASELINE-shaped batch element (config 5 is B=8 over 8 GPUs): weak scaling, no data-path
collective; `--gather` adds the RCCL all_gather of O that a caller wanting the full tensor on every
rank would pay.  Timing: barrier + synchronize on both sides of exactly K steps, max over ranks.

Rank 0 prints ONE JSON line.  `roofline` is the dominant (only) kernel against the dense bf16 MFMA
peak, with the kernel's average launch duration measured by HIP events on the launch stream;
`cpu_baseline` is the reference's CPU path for this op (PyTorch CPU SDPA — the reference has no CPU
kernel, ffpa_attn_interface.py:165-176) timed on this box's host cores on a bounded sample.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

from ffpa_attn_amd import ffpa_attn_func  # noqa: E402
from ffpa_attn_amd.flops import attention_fwd_flops  # noqa: E402

# Dense bf16 MFMA peak of one MI355X, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters":
# 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz ~= 2.5 PFLOP/s (measured 2495 TF with 32x32x16).
MFMA_BF16_PEAK_TFLOPS = 2500.0

WORKLOADS = {
  # name: (B, Hq, Hkv, Nq, Nkv, D, causal)
  "cfg2": (1, 32, 32, 8192, 8192, 512, False),   # BASELINE configs[1] — the headline shape
  "cfg3": (1, 32, 32, 8192, 8192, 1024, False),  # configs[2]
  "cfg2_causal": (1, 32, 32, 8192, 8192, 512, True),
  "cfg4": (2, 32, 8, 8192, 2048, 320, False),    # configs[3] without the mask (see tests for parity)
}


def measured_traffic(workload: str):
  """HBM bytes per launch from the committed rocprofv3 PMC pass of THIS command (profiles/): bench.py
  cannot profile itself, so it reports the figure measured with `rocprofv3 --pmc FETCH_SIZE` (x2: gfx950
  counts 128-B requests at 64 B, MI355X_MICROARCH.md §HBM) + `--pmc WRITE_SIZE`, or null if absent."""
  path = os.path.join(ROOT, "profiles", "r01_bench_cfg2_pmc.json")
  if workload != "cfg2" or not os.path.exists(path):
    return None
  try:
    d = json.load(open(path))["derived"]
    return int(d["hbm_read_bytes_corrected_x2"] + d.get("hbm_write_bytes", 0))
  except (KeyError, ValueError):
    return None


def cpu_baseline(seconds_budget: float = 20.0) -> dict:
  """The reference's CPU path (torch CPU SDPA) on a bounded sample of the same workload: H=4 of the
  32 heads of B=1 N=8192 D=512 bf16, warm-up 1 + best of 3."""
  cores = os.cpu_count() or 1
  torch.set_num_threads(cores)
  H = 4
  torch.manual_seed(0)
  q = torch.randn(1, H, 8192, 512, dtype=torch.bfloat16)
  k = torch.randn(1, H, 8192, 512, dtype=torch.bfloat16)
  v = torch.randn(1, H, 8192, 512, dtype=torch.bfloat16)
  flops = attention_fwd_flops(1, H, 8192, 8192, 512)
  t_begin = time.perf_counter()
  torch._C._nn.scaled_dot_product_attention(q, k, v)  # warm-up
  best = float("inf")
  reps = 0
  while reps < 3 and time.perf_counter() - t_begin < seconds_budget:
    t0 = time.perf_counter()
    torch._C._nn.scaled_dot_product_attention(q, k, v)
    best = min(best, time.perf_counter() - t0)
    reps += 1
  out = {
    "value": round(flops / best / 1e12, 4),
    "unit": "TFLOPS",
    "cores": torch.get_num_threads(),
    "kind": "reference",
    "sample": f"torch CPU SDPA (the reference's CPU path: ffpa_attn_func falls back to it) on B=1 H={H} N=8192 "
              f"D=512 bf16, best of {reps} after 1 warm-up, {best * 1e3:.1f} ms per pass",
  }
  try:  # the oracle (scalar C restatement of the kernel's recurrence), for scale: one thread, 64 rows of one head
    from oracle import ffpa_oracle as fo
    rows = 64
    qb, dname = fo.torch_to_bits(q[:, :1, :rows].contiguous())
    kb, _ = fo.torch_to_bits(k[:, :1].contiguous())
    vb, _ = fo.torch_to_bits(v[:, :1].contiguous())
    t0 = time.perf_counter()
    fo.oracle_forward(qb, kb, vb, dname, scale=512 ** -0.5)
    dt = time.perf_counter() - t0
    out["oracle_port"] = {"value": round(attention_fwd_flops(1, 1, rows, 8192, 512) / dt / 1e12, 6), "unit": "TFLOPS", "cores": 1,
                          "kind": "port", "sample": f"oracle/ffpa_oracle.c on {rows} rows x 8192 keys of one head, D=512, {dt * 1e3:.0f} ms"}
  except Exception as exc:  # the oracle is test infrastructure: its absence must not break the bench line
    out["oracle_port"] = {"error": str(exc)[:120]}
  return out


def main() -> None:
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--warmup", type=int, default=5)
  ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
  ap.add_argument("--gather", action="store_true", help="include an RCCL all_gather of O in the timed step (N > 1)")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-sdpa", action="store_true", help="skip the SDPA-on-GPU comparison")
  args = ap.parse_args()

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if not torch.cuda.is_available():
    sys.exit("bench.py needs a GPU (the HIP kernel has no CPU fallback)")
  torch.cuda.set_device(local_rank)
  dev = torch.device("cuda", local_rank)
  dist = None
  if world > 1:
    import torch.distributed as dist

    dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
  assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

  B, Hq, Hkv, Nq, Nkv, D, causal = WORKLOADS[args.workload]
  torch.manual_seed(0 + rank)  # every rank owns its own batch element(s): weak scaling
  q = torch.randn(B, Hq, Nq, D, dtype=torch.bfloat16, device=dev)
  k = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device=dev)
  v = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device=dev)
  gqa = Hq != Hkv
  flops_per_rank = attention_fwd_flops(B, Hq, Nq, Nkv, D, causal)

  gathered = None
  if args.gather and world > 1:
    gathered = torch.empty((world * B, Hq, Nq, D), dtype=torch.bfloat16, device=dev)

  def step():
    o = ffpa_attn_func(q, k, v, is_causal=causal, enable_gqa=gqa)
    if gathered is not None:
      dist.all_gather_into_tensor(gathered, o)
    return o

  for _ in range(args.warmup):
    step()
  torch.cuda.synchronize()

  # ---- timed region: exactly K steps, barrier + synchronize on both sides -----------------
  starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
  ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
  if dist is not None:
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for i in range(args.steps):
    starts[i].record()  # the kernel is launched on torch's current stream; so are these events
    out = step()
    ends[i].record()
  torch.cuda.synchronize()
  if dist is not None:
    dist.barrier()
  elapsed = time.perf_counter() - t0
  if dist is not None:
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
  kernel_ms = sorted(s.elapsed_time(e) for s, e in zip(starts, ends))
  kernel_ms_avg = sum(kernel_ms) / len(kernel_ms)

  total_flops = flops_per_rank * world * args.steps
  value = total_flops / elapsed / 1e12

  if rank == 0:
    achieved = flops_per_rank / (kernel_ms_avg * 1e-3) / 1e12
    line = {
      "metric": "attention fwd TFLOPS + max-abs-err vs SDPA, bf16 B=1 H=32 N=8192 D=512",
      "value": round(value, 2),
      "unit": "TFLOPS",
      "n_gpus": world,
      "steps": args.steps,
      "warmup": args.warmup,
      "ms_per_step": round(elapsed / args.steps * 1e3, 4),
      "higher_is_better": True,
      "scaling": "weak",
      "vs_baseline": None,  # BASELINE.md holds no published MI355X number for this metric
      "dtype": "bf16",
      "data": "synthetic",
      "config": {
        "workload": f"{args.workload}: B={B} Hq={Hq} Hkv={Hkv} Nq={Nq} Nkv={Nkv} D={D} bf16 "
                    f"{'causal ' if causal else ''}attention forward per GPU (BASELINE configs[1] shape)"
                    if args.workload == "cfg2" else
                    f"{args.workload}: B={B} Hq={Hq} Hkv={Hkv} Nq={Nq} Nkv={Nkv} D={D} bf16 "
                    f"{'causal ' if causal else ''}attention forward per GPU",
        "global_batch": B * world,
        "seq_len": Nq,
        "parallelism": f"(batch,head)-sharded x{world}, no data-path collective" + (" + all_gather(O)" if gathered is not None else ""),
        "flops_model": "4*B*Hq*D*valid_pairs",
      },
      "roofline": {
        "bound": "mfma",
        "achieved": round(achieved, 2),
        "peak": MFMA_BF16_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4),
        "traffic": measured_traffic(args.workload),
        "kernel": "ffpa_fwd_split_d_kernel",
        "kernel_ms_avg": round(kernel_ms_avg, 4),
        "kernel_ms_median": round(kernel_ms[len(kernel_ms) // 2], 4),
        "flops_per_launch": flops_per_rank,
      },
    }
    if world == 1 and not args.no_sdpa:
      # accuracy + the same-device SDPA number the 1.5x target is relative to
      try:
        ref = torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=causal, enable_gqa=gqa)
        line["max_abs_err_vs_sdpa"] = round((out.float() - ref.float()).abs().max().item(), 6)
        for _ in range(2):
          torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=causal, enable_gqa=gqa)
        torch.cuda.synchronize()
        reps = 5
        t1 = time.perf_counter()
        for _ in range(reps):
          torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=causal, enable_gqa=gqa)
        torch.cuda.synchronize()
        sdpa_s = (time.perf_counter() - t1) / reps
        line["sdpa_gpu_tflops"] = round(flops_per_rank / sdpa_s / 1e12, 2)
        line["speedup_vs_sdpa_gpu"] = round((flops_per_rank / (elapsed / args.steps)) / (flops_per_rank / sdpa_s), 3)
      except Exception as e:  # noqa: BLE001 - the comparison is informative, the metric above is not
        line["sdpa_error"] = str(e)[:200]
    if world == 1 and not args.no_cpu_baseline:
      line["cpu_baseline"] = cpu_baseline()
    print(json.dumps(line), flush=True)

  if dist is not None:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
