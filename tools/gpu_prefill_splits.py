"""KV-split count of exactly-filled / under-filled PREFILL launches (developer tool): time hip.forward with num_splits forced to 1 .. 4 next to
the library's own rule (0), interleaved, HIP events.  Cases: `cross` (Nq 1024: 256 workgroups = one per CU), config 4 shapes, a few more."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ffpa_attn_amd import hip

CASES = {
  "cross": (1, 32, 32, 1024, 8192, 512, False), "cross_d320": (1, 32, 32, 1024, 8192, 320, False), "cross_d1024": (1, 32, 32, 1024, 8192, 1024, False),
  "cfg4_nomask": (2, 32, 8, 8192, 2048, 320, False), "n1024": (1, 32, 32, 1024, 1024, 512, False), "n2048": (1, 32, 32, 2048, 2048, 512, False),
  "h8_n4096": (1, 8, 8, 4096, 4096, 512, False), "h12_n4096": (1, 12, 12, 4096, 8192, 512, False),
  # ragged rounds: 1 < workgroups / CUs <= 1.5 (256 CUs, 128-row tiles: 32 row tiles per head at Nq 4096)
  "h9_n4096": (1, 9, 9, 4096, 8192, 512, False), "h10_n4096": (1, 10, 10, 4096, 8192, 512, False), "h11_n4096": (1, 11, 11, 4096, 8192, 512, False),
  "h12_n4096_short": (1, 12, 12, 4096, 2048, 512, False), "h12_n4096_causal": (1, 12, 12, 4096, 4096, 512, True), "h10_d320": (1, 10, 10, 4096, 8192, 320, False),
  "h20_n4096_d1024": (1, 5, 5, 4096, 8192, 1024, False), "h6_d1024": (1, 6, 6, 4096, 8192, 1024, False), "h40_n1024": (1, 40, 40, 1024, 8192, 512, False),
  "h17": (1, 17, 17, 4096, 8192, 512, False), "h20": (1, 20, 20, 4096, 8192, 512, False),
  # under-filled (workgroups <= CUs / 2) with CUs / workgroups far from a whole number
  "h3_n4096": (1, 3, 3, 4096, 8192, 512, False), "h5_n2048": (1, 5, 5, 2048, 8192, 512, False), "h3_n4096_16k": (1, 3, 3, 4096, 16384, 512, False),
  "h7_n1024": (1, 7, 7, 1024, 8192, 512, False), "h3_n2048_d1024": (1, 3, 3, 2048, 8192, 1024, False), "h12_n1024": (1, 12, 12, 1024, 16384, 512, False),
  # part of one round: CUs / 2 < workgroups < CUs
  "h5_n4096": (1, 5, 5, 4096, 8192, 512, False), "h6_n4096": (1, 6, 6, 4096, 8192, 512, False), "h7_n4096": (1, 7, 7, 4096, 8192, 512, False),
  "h5_n4096_short": (1, 5, 5, 4096, 2048, 512, False), "h5_n4096_causal": (1, 5, 5, 4096, 4096, 512, True), "h3_d1024": (1, 3, 3, 4096, 8192, 1024, False),
  "h5_d320": (1, 5, 5, 4096, 8192, 320, False), "h20_n1024": (1, 20, 20, 1024, 8192, 512, False), "h5_n4096_16k": (1, 5, 5, 4096, 16384, 512, False),
  # CAUSAL launches of one round or less (CUs / 2 < workgroups <= CUs) whose longest row tile walks >= 1.5 x the average one: a whole prompt, few heads per GPU
  "c_h8_n4096": (1, 8, 8, 4096, 4096, 512, True), "c_h8_n4096_d128": (1, 8, 8, 4096, 4096, 128, True), "c_h8_n4096_d320": (1, 8, 8, 4096, 4096, 320, True),
  "c_h6_n4096": (1, 6, 6, 4096, 4096, 512, True), "c_b2h8_n2048": (2, 8, 8, 2048, 2048, 512, True), "c_b4h8_n1024": (4, 8, 8, 1024, 1024, 512, True),
  "c_h16_n2048": (1, 16, 16, 2048, 2048, 512, True), "c_h4_n8192": (1, 4, 4, 8192, 8192, 512, True), "c_h4_n4096_d1024": (1, 4, 4, 4096, 4096, 1024, True),
  "c_h32g4_n1024": (1, 32, 8, 1024, 1024, 512, True), "c_h8g4_n4096": (1, 8, 2, 4096, 4096, 512, True), "c_h8_n4096_ctx": (1, 8, 8, 4096, 8192, 512, True),
  "c_h5_n4096": (1, 5, 5, 4096, 4096, 512, True), "c_h7_n4096": (1, 7, 7, 4096, 4096, 512, True), "c_h3_n8192": (1, 3, 3, 8192, 8192, 512, True),
  # UNDER-FILLED causal launches (workgroups <= CUs / 2): the plan's uniform ranges (req 0) next to per-row-tile ranges (TILE_RANGES=1, forced counts)
  "u_h4_n4096": (1, 4, 4, 4096, 4096, 512, True), "u_h2_n4096": (1, 2, 2, 4096, 4096, 512, True), "u_h2_n8192": (1, 2, 2, 8192, 8192, 512, True), "u_h1_n8192": (1, 1, 1, 8192, 8192, 512, True),
  "u_h8_n2048": (1, 8, 8, 2048, 2048, 512, True), "u_h4_n4096_d128": (1, 4, 4, 4096, 4096, 128, True), "u_h2_n4096_d1024": (1, 2, 2, 4096, 4096, 1024, True), "u_h8g4_n2048": (1, 8, 2, 2048, 2048, 512, True),
  "u_h3_n4096": (1, 3, 3, 4096, 4096, 512, True), "u_h4_n2048_ctx": (1, 4, 4, 2048, 8192, 512, True),
  # causal launches of ONE TO TWO rounds (CUs < workgroups < 2 CUs): the longest row tile still outlasts the average CU's share
  "r_h9_n4096": (1, 9, 9, 4096, 4096, 512, True), "r_h10_n4096": (1, 10, 10, 4096, 4096, 512, True), "r_h12_n4096": (1, 12, 12, 4096, 4096, 512, True), "r_h14_n4096": (1, 14, 14, 4096, 4096, 512, True),
  "r_h16_n4096": (1, 16, 16, 4096, 4096, 512, True), "r_h6_n8192": (1, 6, 6, 8192, 8192, 512, True), "r_h8_n8192": (1, 8, 8, 8192, 8192, 512, True), "r_h12_n4096_d320": (1, 12, 12, 4096, 4096, 320, True),
  "r_h3_n8192_d1024": (1, 3, 3, 8192, 8192, 1024, True), "r_h24g4_n2048": (1, 24, 6, 2048, 2048, 512, True), "r_h12_n4096_d128": (1, 12, 12, 4096, 4096, 128, True), "r_b3h4_n4096": (3, 4, 4, 4096, 4096, 512, True),
}
if os.environ.get("ONLY"):
  CASES = {k_: v_ for k_, v_ in CASES.items() if k_ in os.environ["ONLY"].split(",")}
hip.load_library()
# TILE_RANGES=1: forced counts are PER-ROW-TILE ranges of a causal launch (the packed-sequence kernel's dense mode) instead of uniform ranges
FORCE = hip.FLAG_FORCE_SPLITS | (hip.FLAG_TILE_RANGES if os.environ.get("TILE_RANGES") else 0) | int(os.environ.get("EXTRA_FLAGS", "0"), 0)  # (EXTRA_FLAGS: e.g. 0x2 = no XCD remap, 0x400 = a head's row tiles over all eight XCDs)
for name, (B, Hq, Hkv, Nq, Nkv, D, causal) in CASES.items():
  torch.manual_seed(0)
  q = torch.randn(B, Hq, Nq, D, dtype=torch.bfloat16, device="cuda")
  k = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  v = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  flops = 4 * B * Hq * D * Nq * Nkv // (2 if causal else 1)
  arms = [int(x) for x in os.environ.get('ARMS', '0,1,2,3,4').split(',')]
  times = {a: [] for a in arms}
  plans = {}
  # arms 3000 + f (TILE_RANGES=1): ranges of f % of the longest row tile's KV tiles — only longer row tiles split, in two, jobs in descending length (num_splits = 2000 + tiles)
  bc_ = hip.tile_config(hip.padded_head_dim(D))["block_keys"]
  ntv = -(-Nkv // bc_)
  nsp = {a: (a - 5000 if a >= 5000 else (2000 + -(-ntv * (a - 3000) // 100) if a >= 3000 else a)) for a in arms}  # (5000 + n: n ranges merged by the merge KERNEL: no pair fold)
  mil = {a: (False if a >= 5000 else None) for a in arms}
  for a in arms:
    p = {}
    hip.forward(q, k, v, None, causal, D ** -0.5, num_splits=nsp[a], return_lse=False, plan_out=p, flags=FORCE if a > 1 else 0, merge_in_launch=mil[a])
    plans[a] = p.get("splits")
  for _ in range(7):
    for a in arms:
      s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      s.record()
      for _ in range(5):
        hip.forward(q, k, v, None, causal, D ** -0.5, num_splits=nsp[a], return_lse=False, flags=FORCE if a > 1 else 0, merge_in_launch=mil[a])
      e.record()
      torch.cuda.synchronize()
      times[a].append(s.elapsed_time(e) / 5)
  print(f"SPLITS {name:12s} B{B} H{Hq}/{Hkv} Nq{Nq} Nkv{Nkv} D{D}: " + "  ".join(f"req {a} (plan {plans[a]}): {sorted(times[a])[3]:.4f} ms {flops / sorted(times[a])[3] / 1e9:7.1f} TF" for a in arms), flush=True)
