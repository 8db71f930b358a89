"""Interleaved A/B timing of kernel variants on one MI355X (developer tool).

    python tools/gpu_ab.py [--shape B,H,N,D] [--rounds 5] [--reps 5] TAG[:flags] TAG ...

Each TAG is a library built by `python -m ffpa_attn_amd.build --variant TAG DEF...` ("main" = the
shipped ffpa_attn_amd/libffpa_attn_hip.so).  Variants are timed in interleaved rounds inside ONE
process (cdna guide §5.4 rule 24) with HIP events; the report gives median / min ms and TFLOPS per
variant, plus max |O - O_main| to flag variants that changed results (ablations do, by design).
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ffpa_attn_amd import hip  # noqa: E402


def lib_for(tag):
  if tag == "main":
    return hip.load_library()
  return hip.load_library(os.path.join(ROOT, "ffpa_attn_amd", "variants", f"libffpa_attn_hip_{tag}.so"))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("tags", nargs="+")
  ap.add_argument("--shape", default="1,32,8192,512")
  ap.add_argument("--hkv", type=int, default=0)
  ap.add_argument("--nkv", type=int, default=0)
  ap.add_argument("--causal", action="store_true")
  ap.add_argument("--dropout", type=float, default=0.0)
  ap.add_argument("--rounds", type=int, default=5)
  ap.add_argument("--reps", type=int, default=5)
  args = ap.parse_args()
  B, H, N, D = (int(x) for x in args.shape.split(","))
  Hkv = args.hkv or H
  Nkv = args.nkv or N
  torch.manual_seed(0)
  q = torch.randn(B, H, N, D, dtype=torch.bfloat16, device="cuda")
  k = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  v = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  from ffpa_attn_amd.flops import attention_fwd_flops

  flops = attention_fwd_flops(B, H, N, Nkv, D, args.causal)
  variants = []
  for t in args.tags:
    tag, _, fl = t.partition(":")
    variants.append((t, lib_for(tag), int(fl or 0)))

  def run(lib, flags):
    hip._lib = lib
    return hip.forward(q, k, v, None, args.causal, D ** -0.5, flags=flags, return_lse=False, dropout_p=args.dropout, philox_seed=7)[0]

  base = None
  times = {t: [] for t, _, _ in variants}
  diffs = {}
  for t, lib, fl in variants:
    o = run(lib, fl)
    torch.cuda.synchronize()
    if base is None:
      base = o.float()
    diffs[t] = (o.float() - base).abs().max().item()
  for _ in range(args.rounds):
    for t, lib, fl in variants:
      s = torch.cuda.Event(enable_timing=True)
      e = torch.cuda.Event(enable_timing=True)
      s.record()
      for _ in range(args.reps):
        run(lib, fl)
      e.record()
      torch.cuda.synchronize()
      times[t].append(s.elapsed_time(e) / args.reps)
  print(f"shape B={B} H={H}/{Hkv} Nq={N} Nkv={Nkv} D={D} causal={args.causal}  flops={flops:.3e}")
  for t, _, _ in variants:
    ts = sorted(times[t])
    med, mn = ts[len(ts) // 2], ts[0]
    print(f"AB {t:28s} median {med:8.4f} ms {flops / med / 1e9:8.1f} TF | best {mn:8.4f} ms {flops / mn / 1e9:8.1f} TF | maxdiff vs first {diffs[t]:.3e}")


if __name__ == "__main__":
  main()
