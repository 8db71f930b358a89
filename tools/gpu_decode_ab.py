"""A/B of two builds on short-query (decode) shapes (developer tool): python tools/gpu_decode_ab.py TAG_A TAG_B"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffpa_attn_amd import hip
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def lib_for(tag):
  return hip.load_library() if tag == "main" else hip.load_library(os.path.join(ROOT, "ffpa_attn_amd", "variants", f"libffpa_attn_hip_{tag}.so"))
tags = sys.argv[1:] or ["base", "main"]
libs = {t: lib_for(t) for t in tags}
for (B, Hq, Hkv, Nq, Nkv, D) in [(1, 32, 32, 1, 8192, 512), (8, 32, 8, 1, 8192, 512), (1, 32, 8, 1, 8192, 512), (8, 32, 32, 1, 8192, 512), (1, 32, 32, 1, 32768, 512),
                                 (8, 32, 8, 7, 8192, 512), (64, 32, 8, 1, 4096, 512), (8, 32, 8, 1, 8192, 256), (8, 32, 8, 1, 8192, 128), (16, 32, 8, 1, 8192, 1024)]:
  torch.manual_seed(0)
  q = torch.randn(B, Hq, Nq, D, dtype=torch.bfloat16, device="cuda")
  k = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  v = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  bytes_ = (2 * B * Hkv * Nkv * D + 2 * B * Hq * Nq * D) * 2
  outs, res = {}, {}
  for t in tags:
    hip._lib = libs[t]
    plan = {}
    outs[t] = hip.forward(q, k, v, None, False, D ** -0.5, plan_out=plan)[0]
    ts = []
    for rnd in range(5):
      s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      s.record()
      for _ in range(20):
        hip.forward(q, k, v, None, False, D ** -0.5, return_lse=False)
      e.record(); torch.cuda.synchronize()
      ts.append(s.elapsed_time(e) / 20)
    res[t] = (sorted(ts)[2], plan.get("splits"))
  diff = (outs[tags[0]].float() - outs[tags[-1]].float()).abs().max().item()
  print(f"DECODEAB B{B} Hq{Hq}/Hkv{Hkv} Nq{Nq} Nkv{Nkv} D{D}: " + "  ".join(f"{t} {ms * 1e3:.1f} us {bytes_ / ms / 1e9:.2f} TB/s splits {sp}" for t, (ms, sp) in res.items()) + f"  maxdiff {diff:.2e}")
