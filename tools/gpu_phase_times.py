"""Per-phase cycle breakdown of the 16x16x32 prefill kernel (developer tool; needs the instrumented variant:
    python -m ffpa_attn_amd.build --dims 512,1024 --variant timing FFPA_M16_TIMING=1
Every wave writes the shader-clock cycles it spent in six phases of the KV-tile loop over the LSE of its first rows; this script
launches the headline shapes and prints the per-tile means."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ffpa_attn_amd import hip

hip._lib = hip.load_library(os.path.join(ROOT, "ffpa_attn_amd", "variants", "libffpa_attn_hip_timing.so"))
names = ["QK^T loop", "wait A1", "softmax (+K pieces)", "V drain + wait A2", "PV loop", "K drain + wait B"]
for D, BR in ((512, 128), (1024, 64), (320, 128)):
  B, H, N = 1, 32, 8192
  torch.manual_seed(0)
  q, k, v = (torch.randn(B, H, N, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
  for _ in range(3):
    o, lse = hip.forward(q, k, v, None, False, D ** -0.5)
  torch.cuda.synchronize()
  t = lse.view(B, H, N // BR, BR)[..., :32].reshape(-1, 4, 8).double()  # [workgroup, wave, 8]
  tiles = t[..., 7].mean().item()
  per_tile = (t[..., :6].sum(0).sum(0) / (t[..., 7].sum())).tolist()
  total = t[..., 6].mean().item()
  loop = sum(per_tile)
  print(f"PHASE D={D}: {tiles:.0f} KV tiles per workgroup, {loop:.0f} cycles per tile in the loop ({total / tiles:.0f} incl. prologue / epilogue); MFMA-only time per tile: {4096 if D <= 512 else 2048} cycles x (D/512 for D < 512)")
  for n, c in zip(names, per_tile):
    print(f"PHASE   {n:24s} {c:8.0f} cycles  {100 * c / loop:5.1f} %")
