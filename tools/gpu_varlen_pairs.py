"""Paired row tiles per sequence in the packed-sequence kernel (its PAIR build): interleaved same-run A/B against one row tile per workgroup on packed causal
batches — the bench batch, uniform batches of few / many row tiles, ragged mixes, MHA and GQA, four head dims.  Developer tool (tools/visits/)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ffpa_attn_amd import hip  # noqa: E402


def timeit(fn, reps=20, warm=4):
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
  rng = np.random.default_rng(0)
  batches = [
    ("bench batch", [4096, 512, 2048, 1024, 3072, 256, 4864, 512], 32, 8, 512),
    ("bench batch MHA", [4096, 512, 2048, 1024, 3072, 256, 4864, 512], 32, 32, 512),
    ("16 x 1024", [1024] * 16, 32, 8, 512),
    ("8 x 2048", [2048] * 8, 32, 8, 512),
    ("8 x 2048 MHA", [2048] * 8, 32, 32, 512),
    ("4 x 4096", [4096] * 4, 32, 8, 512),
    ("2 x 8192", [8192] * 2, 32, 8, 512),
    ("64 x 256", [256] * 64, 32, 8, 512),
    ("48 ragged 128 ... 2048", [int(x) for x in rng.integers(128, 2048, size=48)], 32, 8, 512),
    ("24 ragged 256 ... 6000", [int(x) for x in rng.integers(256, 6000, size=24)], 32, 8, 512),
    ("24 ragged 256 ... 6000 MHA", [int(x) for x in rng.integers(256, 6000, size=24)], 16, 16, 512),
    ("bench batch D 128", [4096, 512, 2048, 1024, 3072, 256, 4864, 512], 32, 8, 128),
    ("bench batch D 320", [4096, 512, 2048, 1024, 3072, 256, 4864, 512], 32, 8, 320),
    ("bench batch D 1024", [4096, 512, 2048, 1024, 3072, 256, 4864, 512], 16, 4, 1024),
    ("16 x 1024 D 1024", [1024] * 16, 16, 4, 1024),
  ]
  for name, lens, hq, hkv, d in batches:
    t = sum(lens)
    q = torch.randn(t, hq, d, dtype=torch.bfloat16, device="cuda")
    k = torch.randn(t, hkv, d, dtype=torch.bfloat16, device="cuda")
    v = torch.randn(t, hkv, d, dtype=torch.bfloat16, device="cuda")
    cu = torch.tensor([0, *np.cumsum(lens).tolist()], dtype=torch.int32, device="cuda")
    flops = 4 * hq * d * sum(n * (n + 1) // 2 for n in lens)

    def run(flags, plan=None):
      return hip.varlen_forward(q, k, v, cu, cu, max(lens), max(lens), True, d ** -0.5, flags=flags, plan_out=plan)

    pa, pb, pd = {}, {}, {}
    a = run(hip.FLAG_PAIR_TILES, pa)
    b = run(hip.FLAG_NO_PAIR_TILES, pb)
    run(0, pd)
    same = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    ta, tb = [], []
    for _ in range(3):
      tb.append(timeit(lambda: run(hip.FLAG_NO_PAIR_TILES)))
      ta.append(timeit(lambda: run(hip.FLAG_PAIR_TILES)))
    print(f"VARLENPAIR {name:28s} Hq {hq:2d} Hkv {hkv:2d} D {d:4d} row tiles <= {pb['row_tiles']:3d}: one tile per workgroup {min(tb) * 1e3:8.1f} us {flops / min(tb) / 1e9:7.1f} TF ({pb['workgroups']} wgs) | "
          f"paired {min(ta) * 1e3:8.1f} us {flops / min(ta) / 1e9:7.1f} TF ({pa['workgroups']} wgs) | {(min(tb) / min(ta) - 1) * 100:+5.1f} % | bit-identical {same} | default: {'PAIR' if 'PAIR' in pd['kernel'] else 'one tile'}", flush=True)


if __name__ == "__main__":
  main()
