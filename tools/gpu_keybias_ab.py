"""A/B of the LDS key-bias row cache vs per-tile global reads (developer tool): reference bench case 'attn-mask'."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffpa_attn_amd import hip
from ffpa_attn_amd.flops import attention_fwd_flops
for D in (512, 1024, 448):
  torch.manual_seed(0)
  q, k, v = (torch.randn(1, 32, 8192, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
  torch.manual_seed(1)
  bias = torch.randn(1, 1, 1, 8192, dtype=torch.bfloat16, device="cuda") * 0.25
  fl = attention_fwd_flops(1, 32, 8192, 8192, D)
  res = {}
  for name, b, flags in (("none", None, 0), ("lds", bias, 0), ("global", bias, hip.FLAG_NO_BIAS_LDS)):
    ts = []
    for rnd in range(4):
      s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      s.record()
      for _ in range(5):
        hip.forward(q, k, v, b, False, D ** -0.5, flags=flags, return_lse=False)
      e.record(); torch.cuda.synchronize()
      ts.append(s.elapsed_time(e) / 5)
    res[name] = sorted(ts)[len(ts) // 2]
  print("KEYBIAS D=%d: " % D + "  ".join(f"{n} {t:.4f} ms {fl / t / 1e9:.1f} TF" for n, t in res.items()))
