"""Fixed cost per workgroup (developer tool): kernel time vs number of KV tiles at one row tile per CU, linear fit."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffpa_attn_amd import hip
for D, Nq, H in ((512, 1024, 32), (512, 8192, 4), (1024, 512, 32), (320, 1024, 32)):
  pts = []
  for Nkv in (64, 128, 256, 512, 1024, 2048, 4096):
    torch.manual_seed(0)
    q = torch.randn(1, H, Nq, D, dtype=torch.bfloat16, device="cuda")
    k = torch.randn(1, H, Nkv, D, dtype=torch.bfloat16, device="cuda")
    v = torch.randn(1, H, Nkv, D, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
      hip.forward(q, k, v, None, False, D ** -0.5, return_lse=False, num_splits=1)
    ts = []
    for rnd in range(5):
      s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      s.record()
      for _ in range(20):
        hip.forward(q, k, v, None, False, D ** -0.5, return_lse=False, num_splits=1)
      e.record(); torch.cuda.synchronize()
      ts.append(s.elapsed_time(e) / 20 * 1e3)
    pts.append((Nkv, sorted(ts)[2]))
  n = len(pts); sx = sum(p[0] for p in pts[2:]); sy = sum(p[1] for p in pts[2:]); m = len(pts[2:])
  sxx = sum(p[0] ** 2 for p in pts[2:]); sxy = sum(p[0] * p[1] for p in pts[2:])
  slope = (m * sxy - sx * sy) / (m * sxx - sx * sx); icpt = (sy - slope * sx) / m
  print(f"FIXED D={D} Nq={Nq} H={H}: " + " ".join(f"{a}:{b:.1f}us" for a, b in pts) + f" | fit (Nkv>=256): {slope * 64:.2f} us per 64 keys, intercept {icpt:.1f} us")
