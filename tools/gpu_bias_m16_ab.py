"""A/B of the additive-bias and dropout launches on the 16x16x32 build vs the 32x32x16 build (FFPA_FLAG_NO_M16); developer tool."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffpa_attn_amd import hip
from ffpa_attn_amd.flops import attention_fwd_flops

for D in [int(x) for x in os.environ.get("M16_DIMS", "512,320,1024").split(",")]:
  B, H, N = 1, 32, 8192
  torch.manual_seed(0)
  q, k, v = (torch.randn(B, H, N, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
  fl = attention_fwd_flops(B, H, N, N, D)
  cases = {
    "key_bias_bf16": dict(bias=torch.randn(1, 1, 1, N, dtype=torch.bfloat16, device="cuda") * 0.25),
    "key_bias_f32": dict(bias=torch.randn(1, 1, 1, N, dtype=torch.float32, device="cuda") * 0.25),
    "dense_f32": dict(bias=torch.randn(1, 1, N, N, dtype=torch.float32, device="cuda") * 0.25),
    "dense_bf16_nolds": dict(bias=torch.randn(1, 1, N, N, dtype=torch.bfloat16, device="cuda") * 0.25, flags=hip.FLAG_NO_BIAS_LDS),
    "dropout": dict(bias=None, dropout_p=0.1, philox_seed=7, philox_offset=0),
  }
  for name, kw in cases.items():
    bias = kw.pop("bias"); fl0 = kw.pop("flags", 0)
    res, outs = {}, {}
    for tag, flags in (("m32", hip.FLAG_NO_M16), ("m16", 0), ("m32b", hip.FLAG_NO_M16), ("m16b", 0)):
      o, _ = hip.forward(q, k, v, bias, False, D ** -0.5, flags=flags | fl0, kv_bounds=False, **kw)
      outs[tag] = o
      ts = []
      for rnd in range(4):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
          hip.forward(q, k, v, bias, False, D ** -0.5, flags=flags | fl0, return_lse=False, kv_bounds=False, **kw)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 3)
      res[tag] = sorted(ts)[len(ts) // 2]
    diff = (outs["m16"].float() - outs["m32"].float()).abs().max().item()
    print(f"BIASM16 D{D} {name:18s}: " + "  ".join(f"{n} {t:.4f} ms {fl / t / 1e9:.1f}" for n, t in res.items()) + f"  maxdiff {diff:.2e}")
