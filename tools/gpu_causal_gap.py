"""Where the causal launch's gap to the dense one sits (developer tool; one GPU, the shipped library).

Three launches of the same B / H / Nq / D, interleaved, each timed under continuous load:
  dense     Nkv = N                         every workgroup walks N / BC key tiles
  uniform   Nkv = mean visible tiles x BC   NOT causal: every workgroup walks the MEAN number of tiles a causal workgroup walks (the same
                                            number of workgroups, the same total number of KV steps, no masking, no imbalance)
  causal    Nkv = N, tail-aligned           workgroup i walks (i + 1) BR / BC tiles, the last BR / BC of them under the diagonal
time(uniform) against time(dense) x (steps ratio) is the per-workgroup fixed cost (prologue, epilogue, dispatch) amortised over half as many steps;
time(causal) against time(uniform) is everything causal-specific: unequal workgroups (tail of the launch, L2 reuse between co-resident row
tiles of a head) and the masked diagonal tiles.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffpa_attn_amd import hip  # noqa: E402


def timed(fn, ms_budget=120.0):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  fn()
  e.record()
  torch.cuda.synchronize()
  n = max(10, int(ms_budget / max(s.elapsed_time(e), 1e-3)))
  for _ in range(n // 2):  # (continuous load before the timed launches)
    fn()
  s.record()
  for _ in range(n):
    fn()
  e.record()
  torch.cuda.synchronize()
  return s.elapsed_time(e) / n * 1e3  # us


def main():
  B, H, N = 1, 32, 8192
  for D in (512, 320, 1024):
    torch.manual_seed(0)
    q = torch.randn(B, H, N, D, dtype=torch.bfloat16, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.bfloat16, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.bfloat16, device="cuda")
    plan = {}
    hip.forward(q, k, v, None, True, D ** -0.5, return_lse=False, plan_out=plan)
    br, bc = plan["block_rows"], plan["block_keys"]
    nqt = N // br
    steps_causal = sum(-(-(i + 1) * br // bc) for i in range(nqt))  # KV steps of all row tiles of one head
    mean_steps = steps_causal / nqt
    nkv_u = int(round(mean_steps)) * bc
    ku, vu = k[:, :, :nkv_u].contiguous(), v[:, :, :nkv_u].contiguous()
    pu = {}
    hip.forward(q, ku, vu, None, False, D ** -0.5, return_lse=False, plan_out=pu)
    arms = {
      "dense": lambda: hip.forward(q, k, v, None, False, D ** -0.5, return_lse=False),
      "uniform": lambda: hip.forward(q, ku, vu, None, False, D ** -0.5, return_lse=False),
      "causal": lambda: hip.forward(q, k, v, None, True, D ** -0.5, return_lse=False),
    }
    best = {a: [] for a in arms}
    for _ in range(3):
      for a, fn in arms.items():
        best[a].append(timed(fn))
    t = {a: sorted(x)[1] for a, x in best.items()}
    visible = sum(min(N, r + 1) for r in range(N))  # tail-aligned, Nq == Nkv
    tf = {"dense": 4.0 * B * H * D * N * N / t["dense"] / 1e6, "uniform": 4.0 * B * H * D * N * nkv_u / t["uniform"] / 1e6,
          "causal": 4.0 * B * H * D * visible / t["causal"] / 1e6}
    steps_dense = N // bc
    per_step = t["dense"] / steps_dense
    print(f"CAUSALGAP D={D} tile {br}x{bc} (causal plan: {plan['kernel']}; uniform plan splits {pu['splits']} tile {pu['block_rows']}x{pu['block_keys']}): "
          f"dense {t['dense']:.1f} us {tf['dense']:.0f} TF | uniform Nkv={nkv_u} ({mean_steps:.1f} steps/wg) {t['uniform']:.1f} us {tf['uniform']:.0f} TF | "
          f"causal {t['causal']:.1f} us {tf['causal']:.0f} TF")
    print(f"CAUSALGAP D={D}   uniform / (dense x {round(mean_steps) / steps_dense:.4f}) = {t['uniform'] / (t['dense'] * round(mean_steps) / steps_dense):.4f}  (fixed cost per workgroup over half the steps)"
          f" | causal / uniform = {t['causal'] / t['uniform']:.4f}  (imbalance + L2 + diagonal; the diagonal's masked half-tiles alone: counted FLOPs causal / uniform = {visible / (N * nkv_u):.4f})")


if __name__ == "__main__":
  main()
