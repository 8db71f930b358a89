"""A/B of dense-bias handling (developer tool): a [1, H|1, Nq, Nkv] bf16 bias staged through LDS vs read from global memory."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffpa_attn_amd import hip
from ffpa_attn_amd.flops import attention_fwd_flops
for D, hb in ((512, 1), (512, 32), (1024, 1), (448, 1), (320, 1), (256, 1), (128, 1)):
  torch.manual_seed(0)
  N = 8192
  q, k, v = (torch.randn(1, 32, N, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
  bias = torch.randn(1, hb, N, N, dtype=torch.bfloat16, device="cuda") * 0.25
  fl = attention_fwd_flops(1, 32, N, N, D)
  res = {}
  for name, b, flags in (("none", None, 0), ("lds", bias, 0), ("global", bias, hip.FLAG_NO_BIAS_LDS)):
    ts = []
    for rnd in range(4):
      s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      s.record()
      for _ in range(5):
        hip.forward(q, k, v, b, False, D ** -0.5, flags=flags, return_lse=False, kv_bounds=False)
      e.record(); torch.cuda.synchronize()
      ts.append(s.elapsed_time(e) / 5)
    res[name] = sorted(ts)[len(ts) // 2]
  print("DENSEBIAS D=%d bias heads %d: " % (D, hb) + "  ".join(f"{n} {t:.4f} ms {fl / t / 1e9:.1f} TF" for n, t in res.items()))
