"""Decode / short-query bandwidth check on one MI355X (developer tool): the split-KV path is HBM-bound,
algorithmic bytes = K + V read once per KV head (2*Nkv*D*2 B per (batch, kv head)) + Q + O."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ffpa_attn_amd import hip  # noqa: E402

HBM_PEAK = 8.0e12  # B/s spec (MI355X_MICROARCH.md); ~6.3e12 achievable


def run(B, Hq, Hkv, Nq, Nkv, D, reps=20):
  torch.manual_seed(0)
  q = torch.randn(B, Hq, Nq, D, dtype=torch.bfloat16, device="cuda")
  k = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  v = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  plan = {}
  f = lambda: hip.forward(q, k, v, None, False, D ** -0.5, plan_out=plan)  # noqa: E731
  for _ in range(3):
    f()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(reps):
    f()
  e.record()
  torch.cuda.synchronize()
  ms = s.elapsed_time(e) / reps
  sd = lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, enable_gqa=(Hq != Hkv))  # noqa: E731
  try:
    for _ in range(2):
      sd()
    torch.cuda.synchronize()
    s.record()
    for _ in range(5):
      sd()
    e.record()
    torch.cuda.synchronize()
    sdpa_ms = s.elapsed_time(e) / 5
    err = (f()[0].float() - sd().float()).abs().max().item()
  except Exception as ex:  # noqa: BLE001
    sdpa_ms, err = None, str(ex)[:80]
  bytes_ = (2 * B * Hkv * Nkv * D + 2 * B * Hq * Nq * D) * 2
  print("DECODE " + json.dumps({
    "shape": f"B{B} Hq{Hq}/Hkv{Hkv} Nq{Nq} Nkv{Nkv} D{D}", "ms": round(ms, 4), "GBps": round(bytes_ / ms / 1e6, 1),
    "frac_of_8TBps": round(bytes_ / (ms * 1e-3) / HBM_PEAK, 3), "plan": plan, "sdpa_ms": sdpa_ms and round(sdpa_ms, 4),
    "max_abs_vs_sdpa": err}), flush=True)


if __name__ == "__main__":
  for cfg in [(8, 32, 8, 1, 8192, 512), (8, 32, 32, 1, 8192, 512), (1, 32, 8, 1, 8192, 512), (1, 32, 32, 1, 32768, 512),
              (16, 32, 8, 1, 8192, 1024), (8, 32, 8, 7, 8192, 512), (8, 32, 8, 1, 8192, 320), (64, 32, 8, 1, 4096, 512)]:
    run(*cfg)
