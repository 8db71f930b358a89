"""Packed DECODE-like batches through ffpa_attn_varlen_func: one query token per sequence, ragged KV lengths — what continuous batching produces.
The packed kernel runs its 128-row prefill tile on them (one workgroup per (sequence, head) walks that sequence's keys): this prints what that
sustains next to (a) the same batch as a loop of dense decode calls (split-KV kernels, one launch pair per sequence) and (b) the HBM bytes the
batch has to read.  Developer tool (tools/visits/): python tools/gpu_varlen_decode.py"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ffpa_attn_amd import ffpa_attn_varlen_func, hip  # noqa: E402


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
  for nseq, hq, hkv, d, nq in ((32, 32, 32, 512, 1), (32, 32, 8, 512, 1), (8, 32, 32, 512, 1), (64, 32, 8, 320, 1), (16, 32, 8, 512, 16), (32, 32, 8, 512, 4), (64, 64, 8, 128, 8), (16, 16, 2, 1024, 8)):
    lens_k = [int(x) for x in rng.integers(1024, 16384, size=nseq)]
    lens_q = [nq] * nseq
    tq, tk = sum(lens_q), sum(lens_k)
    q = torch.randn(tq, hq, d, dtype=torch.bfloat16, device="cuda")
    k = torch.randn(tk, hkv, d, dtype=torch.bfloat16, device="cuda")
    v = torch.randn(tk, hkv, d, dtype=torch.bfloat16, device="cuda")
    cu_q = torch.tensor([0, *np.cumsum(lens_q).tolist()], dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0, *np.cumsum(lens_k).tolist()], dtype=torch.int32, device="cuda")
    bq, bk = np.cumsum([0, *lens_q]), np.cumsum([0, *lens_k])
    gqa = hq != hkv

    def packed():
      return ffpa_attn_varlen_func(q, k, v, cu_q, cu_k, nq, max(lens_k), causal=True, enable_gqa=gqa)

    def seq(t, a, b):
      return t[a:b].transpose(0, 1).unsqueeze(0)

    def loop():
      # (the op, not ffpa_attn_func: the public entry point sends 8 <= Nq < 512 to SDPA like the reference does — whose is_causal is top-left aligned)
      return [hip.ffpa_attn_forward_hip(seq(q, bq[i], bq[i + 1]), seq(k, bk[i], bk[i + 1]), seq(v, bk[i], bk[i + 1]), None, causal=True, softmax_scale=d ** -0.5)[0] for i in range(nseq)]

    def packed_flags(flags):
      return lambda: hip.varlen_forward(q, k, v, cu_q, cu_k, nq, max(lens_k), True, d ** -0.5, flags=flags)

    o = packed()
    ref = loop()
    err = max((o[bq[i]:bq[i + 1]].float() - ref[i][0].transpose(0, 1).float()).abs().max().item() for i in range(nseq))
    t_p, t_l = timeit(packed), timeit(loop, reps=5, warm=2)
    kv_bytes = 2 * tk * hkv * d * 2
    # the non-temporal K / V fetch, interleaved A/B of the two builds (the default launch above takes one of them by the launch side's rule)
    plan = {}
    hip.varlen_forward(q, k, v, cu_q, cu_k, nq, max(lens_k), True, d ** -0.5, plan_out=plan)
    ab = {"plain": [], "nt": []}
    for _ in range(3):
      ab["plain"].append(timeit(packed_flags(hip.FLAG_NO_KV_STREAM)))
      ab["nt"].append(timeit(packed_flags(hip.FLAG_KV_STREAM)))
    same = all(torch.equal(a, b) for a, b in zip(packed_flags(hip.FLAG_NO_KV_STREAM)(), packed_flags(hip.FLAG_KV_STREAM)()))
    if gqa:
      # the rows of a tile: (head of the KV group, token) packed — or one workgroup per query head (FLAG_NO_PACK_GQA)
      pk = {"packed": [], "per head": []}
      for _ in range(3):
        pk["packed"].append(timeit(packed_flags(0)))
        pk["per head"].append(timeit(packed_flags(hip.FLAG_NO_PACK_GQA)))
      same = all(torch.equal(a, b) for a, b in zip(hip.varlen_forward(q, k, v, cu_q, cu_k, nq, max(lens_k), True, d ** -0.5, num_splits=1),
                                                   hip.varlen_forward(q, k, v, cu_q, cu_k, nq, max(lens_k), True, d ** -0.5, num_splits=1, flags=hip.FLAG_NO_PACK_GQA)))
      print(f"VARLENDECODE_PACK rows = (head of the group, token): {min(pk['packed']) * 1e3:8.1f} us = {kv_bytes / min(pk['packed']) / 1e9:5.2f} TB/s | one workgroup per query head "
            f"{min(pk['per head']) * 1e3:8.1f} us = {kv_bytes / min(pk['per head']) / 1e9:5.2f} TB/s ({min(pk['per head']) / min(pk['packed']):.2f} x) | bit-identical (one KV range) {same}", flush=True)
    print(f"VARLENDECODE_NT default = {plan['kernel']} | plain {min(ab['plain']) * 1e3:8.1f} us = {kv_bytes / min(ab['plain']) / 1e9:5.2f} TB/s | NT {min(ab['nt']) * 1e3:8.1f} us = "
          f"{kv_bytes / min(ab['nt']) / 1e9:5.2f} TB/s ({(min(ab['plain']) / min(ab['nt']) - 1) * 100:+.1f} %) | bit-identical {same}", flush=True)
    print(f"VARLENDECODE {nseq} seqs x Nq {nq}, KV {min(lens_k)} ... {max(lens_k)} (sum {tk}), Hq {hq} Hkv {hkv} D {d}: packed {t_p * 1e3:8.1f} us = {kv_bytes / t_p / 1e9:6.2f} TB/s of K + V | "
          f"loop of {nseq} dense decode calls {t_l * 1e3:8.1f} us = {kv_bytes / t_l / 1e9:6.2f} TB/s | max abs diff {err:.2e}", flush=True)


if __name__ == "__main__":
  main()
