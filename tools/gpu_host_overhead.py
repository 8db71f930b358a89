"""Host time of one decode step (developer tool): how long `ffpa_attn_func` keeps the CPU per call before the kernel is enqueued.

    python tools/gpu_host_overhead.py

Each call is timed with perf_counter WITHOUT synchronising (the call returns once the launch is enqueued; the queue is drained between batches so
that it never fills and blocks), median over 2000 calls, for: the inference path (round 5: no autograd node, no dispatcher round trip, no LSE tensor,
workspace kept per stream, plan scratch asked once per shape class), the same call behind the autograd node (an input requires a gradient: the
registered op + autograd.Function, LSE allocated and saved), and the bare C-ABI launch wrapper `hip.forward`.
"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffpa_attn_amd import ffpa_attn_func, hip  # noqa: E402


def host_us(fn, n=2000, batch=50):
  for _ in range(20):
    fn()
  torch.cuda.synchronize()
  ts = []
  for i in range(n):
    t0 = time.perf_counter()
    fn()
    ts.append(time.perf_counter() - t0)
    if i % batch == batch - 1:
      torch.cuda.synchronize()
  return statistics.median(ts) * 1e6, min(ts) * 1e6


def main():
  torch.manual_seed(0)
  for name, (B, Hq, Hkv, Nkv, D) in {"decode B1 H32 Nkv8192 D512": (1, 32, 32, 8192, 512), "decode B8 GQA 32/8 Nkv8192 D512": (8, 32, 8, 8192, 512)}.items():
    q = torch.randn(B, Hq, 1, D, dtype=torch.bfloat16, device="cuda")
    k = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
    v = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
    qg = q.clone().requires_grad_(True)
    gqa = Hq != Hkv
    from ffpa_attn_amd import DecodeStep

    dstep = DecodeStep(enable_gqa=gqa)
    rows = [
      ("DecodeStep (the call captured once, hipGraphLaunch per step)", lambda: dstep(q, k, v)),
      ("ffpa_attn_func, inference path", lambda: ffpa_attn_func(q, k, v, enable_gqa=gqa)),
      ("ffpa_attn_func, autograd path (q.requires_grad)", lambda: ffpa_attn_func(qg, k, v, enable_gqa=gqa)),
      ("hip.forward (launch wrapper alone, no LSE)", lambda: hip.forward(q, k, v, None, False, D ** -0.5, return_lse=False)),
    ]
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(50):
      ffpa_attn_func(q, k, v, enable_gqa=gqa)
    s.record()
    for _ in range(500):
      ffpa_attn_func(q, k, v, enable_gqa=gqa)
    e.record()
    torch.cuda.synchronize()
    print(f"HOST {name}: GPU time per step under continuous load {s.elapsed_time(e) / 500 * 1e3:.1f} us")
    for label, fn in rows:
      med, best = host_us(fn)
      print(f"HOST   {label:62s} median {med:6.1f} us  best {best:6.1f} us per call")
    # what a caller that synchronises per token pays (host + GPU, nothing overlaps): 200 steps, each followed by a synchronize
    for label, fn in rows[:2]:
      for _ in range(20):
        fn()
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(200):
        fn()
        torch.cuda.synchronize()
      print(f"HOST   {label:62s} {(time.perf_counter() - t0) / 200 * 1e6:6.1f} us per step when the caller synchronises after every step")
    # capture cost: a new key (another KV length) is a new graph
    ts = []
    for i in range(8):
      kk, vv = k[:, :, : Nkv - 64 * (i + 1)], v[:, :, : Nkv - 64 * (i + 1)]
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      dstep(q, kk, vv)
      torch.cuda.synchronize()
      ts.append(time.perf_counter() - t0)
    print(f"HOST   DecodeStep, first call with a new KV length (warm-up call + capture + first replay): median {statistics.median(ts) * 1e3:.2f} ms")


if __name__ == "__main__":
  main()
