"""Interleaved A/B timing of library builds on one MI355X (developer tool; replaces the single-purpose gpu_*_ab.py scripts).

    python tools/gpu_ab.py [--case NAME[,NAME...]] [--shape B,H,N,D ...] [--rounds 5] [--reps 5] TAG[:flags] TAG ...

Each TAG is a library: "main" = the shipped ffpa_attn_amd/libffpa_attn_hip.so, anything else =
ffpa_attn_amd/variants/libffpa_attn_hip_TAG.so (built by `python -m ffpa_attn_amd.build --variant TAG DEF...`, or a saved
build of another commit).  `:flags` ORs ffpa_fwd_params.flags bits into that arm's launches (tool-level bit 0x10000: KV-split launches
keep the separate merge kernel instead of merging inside the launch).  Arms are timed in interleaved
rounds inside ONE process (cdna guide section 5.4 rule 24) with HIP events; per case and arm: median / best ms, TFLOPS (valid
pairs), and max |O - O_first arm| (ablation builds change results by design; A/B arms of one kernel must not).

Cases (`--case`, default cfg2; `--shape` / `--hkv` / `--nkv` / `--causal` / `--dropout` override the "custom" case): the keys of CASES below —
bench.py's workloads (cfg2 cfg3 cfg4_* causal cross gqa non_aligned dropout), additive biases (key_bias dense_bias[_f32|_heads], ..._d128 / _d256 / _d320 / _d1024),
head-dim sweeps (d64 ... d1024, with _causal / _n2048 forms), sequence-length sweeps (n1024 n2048 n16k[_causal] n32k_h8[_d1024] n4k_d1024 n6k_d1024 n12k_d512 n12k_d1024),
other shapes at D = 1024 (cross_d1024 n2048_d1024 gqa_d1024 b4_d1024 d1024_causal) and the short-query launches (decode decode_b8 decode_d128 decode_d1024 decode_long decode_q16).
Launch flags of include/ffpa_attn.h worth an arm of their own: 0x2 no XCD remapping, 0x10 / 0x20 look-ahead touches forced on / off, 0x100 ... 0x400 = 1 / 2 / 4 / 8 XCDs per head.
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


def _c(B, H, Nq, D, hkv=0, nkv=0, causal=False, bias=None, dropout=0.0, offset0=False):
  return dict(B=B, H=H, Hkv=hkv or H, Nq=Nq, Nkv=nkv or Nq, D=D, causal=causal, bias=bias, dropout=dropout, offset0=offset0)


CASES = {
  "cfg2": _c(1, 32, 8192, 512), "cfg3": _c(1, 32, 8192, 1024),
  "cfg4_mask": _c(2, 32, 8192, 320, hkv=8, nkv=2048, bias="tril_bool"), "cfg4_offset0": _c(2, 32, 8192, 320, hkv=8, nkv=2048, causal=True, offset0=True),
  "cfg4_nomask": _c(2, 32, 8192, 320, hkv=8, nkv=2048),
  "causal": _c(1, 32, 8192, 512, causal=True), "cross": _c(1, 32, 1024, 512, nkv=8192), "gqa": _c(1, 32, 8192, 512, hkv=8),
  "non_aligned": _c(1, 8, 8191, 512), "dropout": _c(1, 32, 8192, 512, dropout=0.1),
  "key_bias": _c(1, 32, 8192, 512, bias="key"), "dense_bias": _c(1, 32, 8192, 512, bias="dense"), "dense_bias_f32": _c(1, 32, 8192, 512, bias="dense_f32"),
  "dense_bias_heads": _c(1, 32, 8192, 512, bias="dense_heads"),
  "key_bias_d320": _c(1, 32, 8192, 320, bias="key"), "dense_bias_d320": _c(1, 32, 8192, 320, bias="dense"),
  "key_bias_d1024": _c(1, 32, 8192, 1024, bias="key"), "dense_bias_d1024": _c(1, 32, 8192, 1024, bias="dense"),
  "dropout_d320": _c(1, 32, 8192, 320, dropout=0.1), "dropout_d1024": _c(1, 32, 8192, 1024, dropout=0.1),
  "causal2k": _c(4, 32, 2048, 512, causal=True), "causal12k": _c(1, 32, 12288, 512, causal=True), "causal_b4_4k": _c(4, 32, 4096, 512, causal=True), "causal_gqa": _c(1, 32, 8192, 512, hkv=8, causal=True),
  "causal_h8": _c(1, 8, 8192, 512, causal=True), "causal_cross": _c(1, 32, 4096, 512, nkv=16384, causal=True), "d320_causal4k": _c(2, 32, 4096, 320, causal=True), "d320_causal16k": _c(1, 16, 16384, 320, causal=True),
  "d384_causal": _c(1, 32, 8192, 384, causal=True), "d448_causal": _c(1, 32, 8192, 448, causal=True), "d640_causal": _c(1, 32, 8192, 640, causal=True), "d768_causal": _c(1, 32, 8192, 768, causal=True),
  "d1024_causal4k": _c(2, 32, 4096, 1024, causal=True), "d1024_causal_h8": _c(1, 8, 8192, 1024, causal=True),
  "n1024": _c(1, 32, 1024, 512), "n2048": _c(1, 32, 2048, 512), "causal4k": _c(1, 32, 4096, 512, causal=True),
  "d320": _c(1, 32, 8192, 320), "d320_causal": _c(1, 32, 8192, 320, causal=True), "d320_n2048": _c(4, 32, 2048, 320), "d320_gqa": _c(2, 32, 8192, 320, hkv=8),
  "d320_b3": _c(3, 32, 8192, 320), "d320_n6k": _c(1, 32, 6144, 320), "d256_b3": _c(3, 32, 8192, 256), "d192_b3": _c(3, 32, 8192, 192), "d128_b3": _c(3, 32, 8192, 128),
  "d256_mask": _c(2, 32, 8192, 256, hkv=8, nkv=2048, bias="tril_bool"),
  "cfg3_mask": _c(1, 32, 8192, 1024, bias="tril_bool"), "mask_d1024": _c(2, 32, 8192, 1024, hkv=8, nkv=2048, bias="tril_bool"), "d640_mask": _c(1, 32, 8192, 640, bias="tril_bool"),
  "d768_mask": _c(1, 32, 8192, 768, bias="tril_bool"), "d896_mask": _c(1, 32, 8192, 896, bias="tril_bool"), "d384": _c(1, 32, 8192, 384), "d448": _c(1, 32, 8192, 448), "d640": _c(1, 32, 8192, 640), "d768": _c(1, 32, 8192, 768),
  "d1024_causal": _c(1, 32, 8192, 1024, causal=True),
  "d64": _c(1, 32, 8192, 64), "d64_causal": _c(1, 32, 8192, 64, causal=True), "d64_n2048": _c(4, 32, 2048, 64), "key_bias_d128": _c(1, 32, 8192, 128, bias="key"),
  "dense_bias_d128": _c(1, 32, 8192, 128, bias="dense"), "dropout_d128": _c(1, 32, 8192, 128, dropout=0.1), "mask_d128": _c(2, 32, 8192, 128, hkv=8, nkv=2048, bias="tril_bool"),
  "d128": _c(1, 32, 8192, 128), "d192": _c(1, 32, 8192, 192), "d256": _c(1, 32, 8192, 256), "d128_causal": _c(1, 32, 8192, 128, causal=True),
  "d256_causal": _c(1, 32, 8192, 256, causal=True), "d256_n2048": _c(4, 32, 2048, 256), "key_bias_d256": _c(1, 32, 8192, 256, bias="key"),
  "dense_bias_d256": _c(1, 32, 8192, 256, bias="dense"), "dropout_d256": _c(1, 32, 8192, 256, dropout=0.1), "d128_n2048": _c(4, 32, 2048, 128),
  "d576": _c(1, 32, 8192, 576), "d704": _c(1, 32, 8192, 704), "d832": _c(1, 32, 8192, 832), "d896": _c(1, 32, 8192, 896), "d960": _c(1, 32, 8192, 960),
  "cross_d1024": _c(1, 32, 1024, 1024, nkv=8192), "n2048_d1024": _c(1, 32, 2048, 1024), "gqa_d1024": _c(1, 32, 8192, 1024, hkv=8), "b4_d1024": _c(4, 8, 8192, 1024),
  "decode_d1024": _c(1, 32, 1, 1024, nkv=8192), "decode_d128": _c(4, 32, 1, 128, nkv=8192), "decode_long": _c(1, 8, 1, 512, nkv=65536), "decode_q16": _c(1, 32, 16, 512, nkv=8192),
  "n32k_h8": _c(1, 8, 32768, 512), "n32k_h8_d1024": _c(1, 8, 32768, 1024), "n16k": _c(1, 32, 16384, 512), "n16k_causal": _c(1, 32, 16384, 512, causal=True),
  "n4k_d1024": _c(1, 32, 4096, 1024), "n6k_d1024": _c(1, 32, 6144, 1024), "n12k_d512": _c(1, 32, 12288, 512), "n12k_d1024": _c(1, 16, 12288, 1024),
  "decode": _c(1, 32, 1, 512, nkv=8192), "decode_b8": _c(8, 32, 1, 512, hkv=8, nkv=8192),
}


def make_bias(kind, c, dtype, dev):
  if kind is None:
    return None
  g = torch.Generator(device=dev).manual_seed(1)
  Nq, Nkv, H = c["Nq"], c["Nkv"], c["H"]
  if kind == "tril_bool":
    return torch.ones(Nq, Nkv, dtype=torch.bool, device=dev).tril().view(1, 1, Nq, Nkv)
  if kind == "key":
    return (torch.randn(1, 1, 1, Nkv, device=dev, generator=g) * 0.25).to(dtype)
  if kind == "dense":
    return (torch.randn(1, 1, Nq, Nkv, device=dev, generator=g) * 0.25).to(dtype)
  if kind == "dense_f32":
    return torch.randn(1, 1, Nq, Nkv, device=dev, generator=g) * 0.25
  if kind == "dense_heads":
    return (torch.randn(1, H, Nq, Nkv, device=dev, generator=g) * 0.25).to(dtype)
  raise KeyError(kind)


def flops_of(c):
  from ffpa_attn_amd.flops import attention_fwd_flops

  if c["offset0"] or c["bias"] == "tril_bool":
    return 4 * c["B"] * c["H"] * c["D"] * sum(min(c["Nkv"], r + 1) for r in range(c["Nq"]))
  return attention_fwd_flops(c["B"], c["H"], c["Nq"], c["Nkv"], c["D"], c["causal"])


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("tags", nargs="+")
  ap.add_argument("--case", default="")
  ap.add_argument("--shape", default="")
  ap.add_argument("--hkv", type=int, default=0)
  ap.add_argument("--nkv", type=int, default=0)
  ap.add_argument("--causal", action="store_true")
  ap.add_argument("--dropout", type=float, default=0.0)
  ap.add_argument("--rounds", type=int, default=5)
  ap.add_argument("--reps", type=int, default=5)
  args = ap.parse_args()
  cases = []
  if args.shape:
    B, H, N, D = (int(x) for x in args.shape.split(","))
    cases.append(("custom", _c(B, H, N, D, hkv=args.hkv, nkv=args.nkv, causal=args.causal, dropout=args.dropout)))
  for name in [n for n in args.case.split(",") if n]:
    cases.append((name, CASES[name]))
  if not cases:
    cases.append(("cfg2", CASES["cfg2"]))
  variants = []
  for t in args.tags:
    tag, _, fl = t.partition(":")
    variants.append((t, lib_for(tag), int(fl, 0) if fl else 0))

  for cname, c in cases:
    torch.manual_seed(0)
    dt = torch.bfloat16
    q = torch.randn(c["B"], c["H"], c["Nq"], c["D"], dtype=dt, device="cuda")
    k = torch.randn(c["B"], c["Hkv"], c["Nkv"], c["D"], dtype=dt, device="cuda")
    v = torch.randn(c["B"], c["Hkv"], c["Nkv"], c["D"], dtype=dt, device="cuda")
    bias = make_bias(c["bias"], c, dt, "cuda")
    flops = flops_of(c)
    kw = dict(flags=0, return_lse=False, dropout_p=c["dropout"], philox_seed=7)
    if c["offset0"]:
      kw["causal_offset"] = 0

    def run(lib, flags):
      hip._lib = lib
      kw["flags"] = flags & ~0x10000  # (every library flag; bit 0x10000 is this tool's own)
      kw["merge_in_launch"] = not (flags & 0x10000)
      return hip.forward(q, k, v, bias, c["causal"], c["D"] ** -0.5, **kw)[0]

    base = None
    times = {t: [] for t, _, _ in variants}
    diffs = {}
    for t, lib, fl in variants:
      o = run(lib, fl)
      torch.cuda.synchronize()
      if base is None:
        base = o.float()
      diffs[t] = (torch.nan_to_num(o.float()) - torch.nan_to_num(base)).abs().max().item()
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
    print(f"CASE {cname}: B={c['B']} H={c['H']}/{c['Hkv']} Nq={c['Nq']} Nkv={c['Nkv']} D={c['D']} causal={c['causal']} bias={c['bias']} dropout={c['dropout']}  flops={flops:.3e}", flush=True)
    for t, _, _ in variants:
      ts = sorted(times[t])
      med, mn = ts[len(ts) // 2], ts[0]
      print(f"AB {cname:18s} {t:24s} median {med:9.4f} ms {flops / med / 1e9:8.1f} TF | best {mn:9.4f} ms {flops / mn / 1e9:8.1f} TF | maxdiff vs first {diffs[t]:.3e}", flush=True)
    del q, k, v, bias, base


if __name__ == "__main__":
  main()
