"""Kernel-level timing of one short-query launch (developer tool; run under rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffpa_attn_amd import hip
B, Hq, Hkv, Nkv, D = (int(x) for x in (sys.argv[1:6] or (1, 32, 8, 8192, 512)))
q = torch.randn(B, Hq, 1, D, dtype=torch.bfloat16, device="cuda")
k = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
v = torch.randn_like(k)
plan = {}
for _ in range(30):
  hip.forward(q, k, v, None, False, D ** -0.5, return_lse=False, plan_out=plan)
torch.cuda.synchronize()
print("PLAN", plan)
