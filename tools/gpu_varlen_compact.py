"""The compact grid of ragged packed PREFILL batches (grid sized by total_q instead of batch x max_seqlen_q): interleaved A/B against FLAG_NO_COMPACT_GRID (same order,
same bits).  Developer tool (tools/visits/): python tools/gpu_varlen_compact.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ffpa_attn_amd import hip  # noqa: E402


def main():
  torch.manual_seed(0)
  rng = np.random.default_rng(0)
  cases = [  # Hq, Hkv, D, lens
    (32, 8, 512, [4096, 512, 2048, 1024, 3072, 256, 4864, 512]),  # bench.py --workload varlen
    (32, 8, 512, [16384] + [256] * 63),  # one long prompt among short ones
    (32, 8, 512, [8192] + [int(x) for x in rng.integers(64, 1024, size=31)]),
    (32, 32, 512, [int(x) for x in rng.integers(128, 4096, size=16)]),
    (32, 8, 320, [4096, 512, 2048, 1024, 3072, 256, 4864, 512]),
    (16, 2, 1024, [4096, 512, 2048, 1024, 3072, 256, 4864, 512]),
    (32, 8, 128, [8192] + [128] * 127),
    (32, 8, 512, [2048] * 8),  # equal lengths: the full grid stays
  ]
  for hq, hkv, d, lens in cases:
    t = sum(lens)
    q = torch.randn(t, hq, d, dtype=torch.bfloat16, device="cuda")
    k = torch.randn(t, hkv, d, dtype=torch.bfloat16, device="cuda")
    v = torch.randn(t, hkv, d, dtype=torch.bfloat16, device="cuda")
    cu = torch.tensor([0, *np.cumsum(lens).tolist()], dtype=torch.int32, device="cuda")
    flop = 4.0 * hq * d * sum(n * (n + 1) // 2 for n in lens)
    arms = {"compact": 0, "full": hip.FLAG_NO_COMPACT_GRID}
    plans, times = {}, {a: [] for a in arms}
    outs = {}
    for a, f in arms.items():
      p = {}
      outs[a] = hip.varlen_forward(q, k, v, cu, cu, max(lens), max(lens), True, d ** -0.5, flags=f, plan_out=p)
      plans[a] = p["workgroups"]
    same = torch.equal(outs["compact"][0], outs["full"][0]) and torch.equal(outs["compact"][1], outs["full"][1])
    for _ in range(7):
      for a, f in arms.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
          hip.varlen_forward(q, k, v, cu, cu, max(lens), max(lens), True, d ** -0.5, flags=f)
        e1.record()
        torch.cuda.synchronize()
        times[a].append(e0.elapsed_time(e1) / 5)
    med = {a: sorted(ts)[3] for a, ts in times.items()}
    print(f"COMPACT Hq {hq} Hkv {hkv} D {d} {len(lens)} seqs {min(lens)} ... {max(lens)} ({t} tokens): full grid {plans['full']} wgs {med['full'] * 1e3:7.1f} us {flop / med['full'] / 1e9:6.0f} TF | "
          f"compact {plans['compact']} wgs {med['compact'] * 1e3:7.1f} us {flop / med['compact'] / 1e9:6.0f} TF | {med['full'] / med['compact']:.3f} x | bits equal {same}", flush=True)


if __name__ == "__main__":
  main()
