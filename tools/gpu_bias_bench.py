"""Cost of the additive-bias path vs the structured causal path (developer tool)."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ffpa_attn_amd import hip
from ffpa_attn_amd.flops import attention_fwd_flops

def t(fn, reps=10):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(reps): fn()
  e.record(); torch.cuda.synchronize()
  return s.elapsed_time(e) / reps

torch.manual_seed(0)
B, Hq, Hkv, Nq, Nkv, D = 2, 32, 8, 8192, 2048, 320
q = torch.randn(B, Hq, Nq, D, dtype=torch.bfloat16, device="cuda")
k = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
v = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
sc = D ** -0.5
mask = torch.ones(Nq, Nkv, dtype=torch.bool, device="cuda").tril()
bias16 = torch.zeros(1, 1, Nq, Nkv, dtype=q.dtype, device="cuda").masked_fill(~mask, float("-inf"))
bias32 = bias16.float()
kp = torch.zeros(1, 1, 1, Nkv, dtype=q.dtype, device="cuda")
fl_c = attention_fwd_flops(B, Hq, Nq, Nkv, D, True, causal_offset=0)
fl = attention_fwd_flops(B, Hq, Nq, Nkv, D)
res = {}
res["cfg4 structured causal_offset=0"] = (t(lambda: hip.forward(q, k, v, None, True, sc, causal_offset=0)), fl_c)
res["cfg4 explicit bf16 mask [1,1,Nq,Nkv]"] = (t(lambda: hip.forward(q, k, v, bias16, False, sc)), fl_c)
res["cfg4 explicit fp32 mask"] = (t(lambda: hip.forward(q, k, v, bias32, False, sc)), fl_c)
res["cfg4 no mask"] = (t(lambda: hip.forward(q, k, v, None, False, sc)), fl)
res["cfg4 key-padding bias [1,1,1,Nkv]"] = (t(lambda: hip.forward(q, k, v, kp, False, sc)), fl)
res["cfg4 SDPA is_causal"] = (t(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True), 3), fl_c)
for name, (ms, f) in res.items():
  print("BIAS " + json.dumps({"case": name, "ms": round(ms, 4), "tflops": round(f / ms / 1e9, 1)}), flush=True)

# ---- mask-derived tile clipping (kv_bounds): explicit causal mask at config-2 size, and config 4
for (Bx, Hqx, Hkvx, Nqx, Nkvx, Dx) in ((1, 32, 32, 8192, 8192, 512), (2, 32, 8, 8192, 2048, 320)):
  qx = torch.randn(Bx, Hqx, Nqx, Dx, dtype=torch.bfloat16, device="cuda")
  kx = torch.randn(Bx, Hkvx, Nkvx, Dx, dtype=torch.bfloat16, device="cuda")
  vx = torch.randn_like(kx)
  mx = torch.ones(Nqx, Nkvx, dtype=torch.bool, device="cuda").tril()
  bx = torch.zeros(1, 1, Nqx, Nkvx, dtype=torch.bfloat16, device="cuda").masked_fill(~mx, float("-inf"))
  flc = attention_fwd_flops(Bx, Hqx, Nqx, Nkvx, Dx, True, causal_offset=0)
  bounds = hip.mask_kv_bounds(bx, Nqx, Nkvx)
  cases = {
      "mask, every tile": lambda: hip.forward(qx, kx, vx, bx, False, Dx ** -0.5, kv_bounds=False),
      "mask, clipped (bounds precomputed)": lambda: hip.forward(qx, kx, vx, bx, False, Dx ** -0.5, kv_bounds=bounds),
      "mask, clipped (bounds derived per call)": lambda: hip.forward(qx, kx, vx, bx, False, Dx ** -0.5, kv_bounds=True),
      "structured causal_offset=0": lambda: hip.forward(qx, kx, vx, None, True, Dx ** -0.5, causal_offset=0),
  }
  for name, fn in cases.items():
    ms = t(fn)
    print("BIAS " + json.dumps({"case": f"B{Bx} Hq{Hqx}/Hkv{Hkvx} Nq{Nqx} Nkv{Nkvx} D{Dx} tril: {name}", "ms": round(ms, 4), "tflops": round(flc / ms / 1e9, 1)}), flush=True)
