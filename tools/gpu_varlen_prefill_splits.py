"""KV splits inside the packed-sequence launch for PREFILL batches that leave most of the chip idle (several row tiles per head, few (sequence, head, row tile)
workgroups: chunked prefill of a long sequence with few heads per GPU, a tensor-parallel shard of a small batch).  For each batch: the one-range launch, forced
split counts (FLAG_FORCE_SPLITS), what the library picks by itself, and the loop of dense calls (whose plan splits under-filled launches too); TFLOPS of the visible
scores; the worst difference against the one-range launch.  Developer tool (tools/visits/): python tools/gpu_varlen_prefill_splits.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ffpa_attn_amd import hip  # noqa: E402


def timeit(fn, reps=20, warm=3):
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
  cases = [  # Hq, Hkv, D, causal, [(Nq, Nkv) per sequence]
    (8, 8, 512, True, [(1024, 16384)]),  # a 1024-token chunk of a 16k prompt, 8 heads of a TP-8 shard: 64 workgroups
    (8, 8, 512, True, [(4096, 4096)]),  # a whole 4k prompt, 8 heads: 256 row tiles, the short ones finish early
    (8, 8, 512, True, [(2048, 8192)]),
    (4, 4, 512, True, [(2048, 32768)]),
    (4, 1, 512, True, [(512, 65536)]),
    (8, 2, 512, True, [(1024, 16384), (256, 4096)]),
    (2, 2, 512, False, [(1024, 8192)]),
    (8, 8, 512, False, [(512, 8192)]),
    (8, 8, 320, True, [(1024, 16384)]),
    (8, 8, 128, True, [(1024, 16384)]),
    (8, 8, 128, True, [(4096, 4096)]),
    (4, 4, 1024, True, [(512, 16384)]),
    (4, 4, 1024, True, [(2048, 2048)]),
    (16, 16, 512, True, [(512, 16384)]),
    (16, 16, 512, True, [(1024, 8192)]),  # 128 workgroups: exactly half the chip
    (32, 8, 512, True, [(512, 4096)]),
    (32, 8, 512, True, [(256, 32768)]),
  ]
  if os.environ.get("CASES") == "one_round":
    # causal launches of one round or less (CUs / 2 < workgroups <= CUs): what the DENSE call would get from this kernel's per-row-tile ranges (tools/gpu_prefill_splits.py c_*: the dense plan's side)
    cases = [
      (8, 8, 512, True, [(4096, 4096)]), (8, 8, 128, True, [(4096, 4096)]), (8, 8, 320, True, [(4096, 4096)]), (6, 6, 512, True, [(4096, 4096)]), (8, 8, 512, True, [(2048, 2048)] * 2),
      (8, 8, 512, True, [(1024, 1024)] * 4), (16, 16, 512, True, [(2048, 2048)]), (4, 4, 512, True, [(8192, 8192)]), (4, 4, 1024, True, [(4096, 4096)]), (32, 8, 512, True, [(1024, 1024)]),
      (8, 2, 512, True, [(4096, 4096)]), (8, 8, 512, True, [(4096, 8192)]), (5, 5, 512, True, [(4096, 4096)]), (7, 7, 512, True, [(4096, 4096)]), (3, 3, 512, True, [(8192, 8192)]),
      (8, 8, 256, True, [(4096, 4096)]), (8, 8, 192, True, [(4096, 4096)]), (2, 2, 1024, True, [(8192, 8192)]), (8, 8, 512, True, [(3000, 3000)]),
    ]
  for hq, hkv, d, causal, seqs in cases:
    lens_q, lens_k = [a for a, _ in seqs], [b for _, b in seqs]
    tq, tk = sum(lens_q), sum(lens_k)
    q = torch.randn(tq, hq, d, dtype=torch.bfloat16, device="cuda")
    k = torch.randn(tk, hkv, d, dtype=torch.bfloat16, device="cuda")
    v = torch.randn(tk, hkv, d, dtype=torch.bfloat16, device="cuda")
    cu_q = torch.tensor([0, *np.cumsum(lens_q).tolist()], dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0, *np.cumsum(lens_k).tolist()], dtype=torch.int32, device="cuda")
    # visible scores x 4 D FLOP
    vis = sum((nq * nk - nq * (nq - 1) // 2) if causal else nq * nk for nq, nk in seqs)
    flop = 4.0 * hq * vis * d

    def run(splits, flags=0, plan=None):
      return hip.varlen_forward(q, k, v, cu_q, cu_k, max(lens_q), max(lens_k), causal, d ** -0.5, num_splits=splits, flags=flags, plan_out=plan)

    ref, ref_lse = run(1)
    plan = {}
    run(0, plan=plan)
    row = []
    best = (None, 1e9)
    for s in ((1, 2, 3, 4, 6) if os.environ.get("CASES") == "one_round" else (1, 2, 3, 4, 6, 8, 12, 16, 32)):
      pl = {}
      o, l = run(s, hip.FLAG_FORCE_SPLITS if s > 1 else 0, pl)
      if pl["splits"] != s:
        continue
      t = min(timeit(lambda: run(s, hip.FLAG_FORCE_SPLITS if s > 1 else 0)) for _ in range(2))
      err = (o.float() - ref.float()).abs().max().item()
      row.append(f"{s}: {t * 1e3:6.1f} us {flop / t / 1e9:5.0f} TF ({err:.1e})")
      if t < best[1]:
        best = (s, t)
    t_auto = min(timeit(lambda: run(0)) for _ in range(2))

    # the same batch as a loop of dense calls (their plan: paired / split / wide tiles)
    def dense_loop():
      qs = ks = 0
      for nq, nk in seqs:
        hip.forward(q[qs:qs + nq].transpose(0, 1).unsqueeze(0), k[ks:ks + nk].transpose(0, 1).unsqueeze(0), v[ks:ks + nk].transpose(0, 1).unsqueeze(0), None, causal, d ** -0.5,
                    return_lse=False, causal_offset=nk - nq if causal else 0)
        qs, ks = qs + nq, ks + nk

    try:
      t_dense = min(timeit(dense_loop) for _ in range(2))
    except Exception as e:  # noqa: BLE001
      print("dense loop failed:", e)
      t_dense = float("nan")
    t_one = float(row[0].split(":")[1].split("us")[0])
    print(f"VARLENPREFILLSPLITS Hq {hq} Hkv {hkv} D {d} {'causal' if causal else 'full'} {seqs}, {plan['workgroups'] // plan['splits']} workgroups | library: {plan['splits']} ranges "
          f"{t_auto * 1e3:6.1f} us = {flop / t_auto / 1e9:5.0f} TF | best forced {best[0]}: {best[1] * 1e3:6.1f} us | one range / library = {t_one / (t_auto * 1e3):.2f} x | dense loop {t_dense * 1e3:6.1f} us\n    "
          + " | ".join(row), flush=True)


if __name__ == "__main__":
  main()
