"""KV splits inside the packed-sequence launch (ABI 6): decode-like batches of FEW long sequences — (sequence, head) pairs that leave most of the chip idle.
For each batch: the one-range launch, forced split counts (FLAG_FORCE_SPLITS), and what the library picks by itself; HBM rate of K + V; the worst
difference against the one-range launch.  Developer tool (tools/visits/): python tools/gpu_varlen_splits.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ffpa_attn_amd import hip  # noqa: E402


def timeit(fn, reps=30, warm=5):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps


def main():
  torch.manual_seed(0)
  rng = np.random.default_rng(0)
  cases = [  # sequences, Hq, Hkv, D, tokens per sequence, KV length range
    (1, 32, 8, 512, 1, (32768, 32769)), (4, 32, 8, 512, 1, (2048, 16384)), (8, 32, 8, 512, 1, (2048, 16384)), (16, 32, 8, 512, 1, (2048, 16384)),
    (32, 32, 8, 512, 1, (1024, 16384)), (64, 32, 8, 512, 1, (1024, 16384)), (8, 32, 32, 512, 1, (2048, 16384)), (4, 32, 32, 512, 1, (2048, 16384)),
    (8, 32, 8, 320, 1, (2048, 16384)), (8, 64, 8, 128, 1, (2048, 16384)), (8, 16, 2, 1024, 1, (2048, 16384)), (4, 32, 8, 512, 16, (4096, 16384)),
    (8, 8, 8, 512, 64, (4096, 16384)),
  ]
  for nseq, hq, hkv, d, nq, (lo, hi) in cases:
    lens_k = [int(x) for x in rng.integers(lo, hi, size=nseq)]
    lens_q = [nq] * nseq
    tq, tk = sum(lens_q), sum(lens_k)
    q = torch.randn(tq, hq, d, dtype=torch.bfloat16, device="cuda")
    k = torch.randn(tk, hkv, d, dtype=torch.bfloat16, device="cuda")
    v = torch.randn(tk, hkv, d, dtype=torch.bfloat16, device="cuda")
    cu_q = torch.tensor([0, *np.cumsum(lens_q).tolist()], dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0, *np.cumsum(lens_k).tolist()], dtype=torch.int32, device="cuda")
    kv_bytes = 2 * tk * hkv * d * 2

    def run(splits, flags=0, plan=None):
      return hip.varlen_forward(q, k, v, cu_q, cu_k, nq, max(lens_k), True, d ** -0.5, num_splits=splits, flags=flags, plan_out=plan)

    ref, ref_lse = run(1)
    plan = {}
    run(0, plan=plan)
    row = []
    best = (None, 1e9)
    for s in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64):
      pl = {}
      o, l = run(s, hip.FLAG_FORCE_SPLITS if s > 1 else 0, pl)
      if pl["splits"] != s:
        continue
      t = min(timeit(lambda: run(s, hip.FLAG_FORCE_SPLITS if s > 1 else 0)) for _ in range(2))
      err = (o.float() - ref.float()).abs().max().item()
      row.append(f"{s}: {t * 1e3:6.1f} us {kv_bytes / t / 1e9:4.2f} TB/s ({err:.1e})")
      if t < best[1]:
        best = (s, t)
    t_auto = min(timeit(lambda: run(0)) for _ in range(2))
    print(f"VARLENSPLITS {nseq} seqs x Nq {nq}, KV {min(lens_k)} ... {max(lens_k)} (sum {tk}), Hq {hq} Hkv {hkv} D {d}, {plan['workgroups'] // plan['splits']} pairs | library: {plan['splits']} ranges "
          f"{t_auto * 1e3:6.1f} us = {kv_bytes / t_auto / 1e9:4.2f} TB/s | best forced {best[0]}: {best[1] * 1e3:6.1f} us | one range / library = {float(row[0].split(':')[1].split('us')[0]) / (t_auto * 1e3):.2f} x\\n    "
          + " | ".join(row), flush=True)


if __name__ == "__main__":
  main()
