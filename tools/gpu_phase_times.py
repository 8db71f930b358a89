"""Per-phase cycle breakdown of the 16x16x32 prefill kernel (developer tool; needs instrumented variants:
    python -m ffpa_attn_amd.build --dims 512,1024 --variant timing FFPA_M16_TIMING=1 [more -D switches]
Every wave writes the shader-clock cycles it spent in the phases of its KV-tile loop over the LSE of its first rows; this script
launches B=1 H=32 N=8192 at the given head dims and prints the per-tile means.

    python tools/gpu_phase_times.py [--dims 512,1024,320] TAG [TAG ...]        (TAG: variants/libffpa_attn_hip_TAG.so, default "timing")
"""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ffpa_attn_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("tags", nargs="*", default=["timing"])
ap.add_argument("--dims", default="512,1024,320")
args = ap.parse_args()
names = ["QK^T loop (2-half K: key block 1)", "wait A1", "softmax (+pieces)", "V drain + wait A2", "PV loop", "K drain + wait B", "QK^T key block 0", "K2 drain + wait M"]
for tag in args.tags:
  hip._lib = hip.load_library(os.path.join(ROOT, "ffpa_attn_amd", "variants", f"libffpa_attn_hip_{tag}.so"))
  for D in (int(x) for x in args.dims.split(",")):
    BR = 128 if D <= 512 else 64
    B, H, N = 1, 32, 8192
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    for _ in range(3):
      o, lse = hip.forward(q, k, v, None, False, D ** -0.5)
    torch.cuda.synchronize()
    t = lse.view(B, H, N // BR, BR)[..., :64].reshape(-1, 4, 16).double()  # [workgroup, wave, 16]
    tiles = t[..., 7].mean().item()
    cols = [0, 1, 2, 3, 4, 5, 8, 9]
    per_tile = (t[..., cols].sum(0).sum(0) / (t[..., 7].sum())).tolist()
    total = t[..., 6].mean().item()
    loop = sum(per_tile)
    mfma = (4096 if D <= 512 else 2048) * (D / 512 if D < 512 else 1) * (D / 1024 if D > 512 else 1)
    print(f"PHASE {tag} D={D}: {tiles:.0f} KV tiles per workgroup, {loop:.0f} cycles per tile in the loop ({total / tiles:.0f} incl. prologue / epilogue); MFMA-only time per tile: {mfma:.0f} cycles")
    for n, c in zip(names, per_tile):
      if c > 0:
        print(f"PHASE   {n:34s} {c:8.0f} cycles  {100 * c / loop:5.1f} %")
