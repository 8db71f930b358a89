"""Short-query step time vs the number of KV splits (developer tool; separate merge kernel): for each shape the splits that give 1, 1.5, 2 and 3 workgroups
per CU (256 CUs) next to what the library picks by itself."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffpa_attn_amd import hip
import os as _os
SHAPES = [(4, 32, 32, 1, 8192, 128), (2, 32, 32, 1, 16384, 128), (1, 64, 64, 1, 8192, 256), (1, 32, 32, 1, 8192, 192), (1, 32, 32, 1, 8192, 256), (1, 32, 32, 1, 8192, 384), (1, 32, 32, 1, 8192, 64), (1, 32, 32, 1, 8192, 448), (1, 32, 32, 1, 8192, 640), (1, 32, 32, 1, 8192, 768)] if _os.environ.get('SMALL_D') else [(1, 32, 32, 1, 8192, 512), (1, 32, 32, 1, 8192, 1024), (8, 32, 8, 1, 8192, 512), (1, 8, 8, 1, 65536, 512), (4, 32, 32, 1, 8192, 128), (1, 32, 32, 1, 2048, 512),
          (1, 32, 32, 1, 4096, 512), (1, 32, 32, 1, 32768, 512), (1, 32, 32, 16, 8192, 512), (1, 32, 8, 1, 8192, 512), (2, 16, 16, 1, 16384, 320), (16, 32, 32, 1, 4096, 512),
          (1, 64, 64, 1, 8192, 256), (3, 20, 20, 1, 10000, 512)]
for (B, H, Hkv, Nq, Nkv, D) in SHAPES:
  torch.manual_seed(0)
  q = torch.randn(B, H, Nq, D, dtype=torch.bfloat16, device="cuda")
  k = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  v = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  plan = {}
  hip.forward(q, k, v, None, False, D ** -0.5, return_lse=False, plan_out=plan, num_splits=1)
  base = None
  out = []
  tried = set()
  cands = [0]
  # workgroups without splitting = what one split launches (the plan does not expose it: derive from a 2-split plan's workspace? use B * Hkv-packed heads)
  group = H // Hkv
  rows = Nq * group
  base = B * (Hkv if rows <= 32 else H) * ((Nq * (group if rows <= 32 else 1) + 31) // 32)
  for per_cu in ((1.0, 2.0, 3.0, 4.0, 6.0) if _os.environ.get('SMALL_D') else (1.0, 1.5, 2.0, 3.0)):
    cands.append(max(1, int(round(256 * per_cu / base))))
  if _os.environ.get('AUTO_ONLY'):
    cands = [0]
  for ns in cands:
    if ns in tried:
      continue
    tried.add(ns)
    f = lambda: hip.forward(q, k, v, None, False, D ** -0.5, return_lse=False, plan_out=plan, **({"num_splits": ns} if ns else {}))
    for _ in range(5):
      f()
    ts = []
    for _ in range(7):
      s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      s.record()
      for _ in range(20):
        f()
      e.record(); torch.cuda.synchronize()
      ts.append(s.elapsed_time(e) / 20 * 1e3)
    out.append(f"{'auto' if ns == 0 else ns}->{plan['splits']}: {sorted(ts)[3]:.1f}us")
  print(f"SPLITS B{B} H{H}/{Hkv} Nq{Nq} Nkv{Nkv} D{D} (base {base}): " + "  ".join(out), flush=True)
