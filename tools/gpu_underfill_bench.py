"""Under-filled prefill launches: KV-split plan vs one workgroup per row tile (developer tool)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for (B, Hq, Hkv, Nq, Nkv, D, causal) in ((1, 8, 8, 512, 65536, 512, False), (1, 8, 2, 2048, 32768, 512, True), (1, 4, 4, 4096, 4096, 512, False),
                                         (1, 8, 8, 1024, 16384, 320, True), (2, 4, 4, 512, 32768, 1024, False)):
  q = torch.randn(B, Hq, Nq, D, dtype=torch.bfloat16, device="cuda")
  k = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  v = torch.randn_like(k)
  plan = {}
  hip.forward(q, k, v, None, causal, D ** -0.5, plan_out=plan)
  ms_split = t(lambda: hip.forward(q, k, v, None, causal, D ** -0.5, return_lse=False))
  ms_one = t(lambda: hip.forward(q, k, v, None, causal, D ** -0.5, return_lse=False, num_splits=1))
  fl = attention_fwd_flops(B, Hq, Nq, Nkv, D, causal)
  print("UNDERFILL " + json.dumps({"shape": f"B{B} Hq{Hq}/Hkv{Hkv} Nq{Nq} Nkv{Nkv} D{D} causal={causal}", "splits": plan["splits"],
                                   "ms_split": round(ms_split, 4), "ms_unsplit": round(ms_one, 4), "speedup": round(ms_one / ms_split, 2),
                                   "tflops_split": round(fl / ms_split / 1e9, 1)}), flush=True)
