"""What walking the later row tiles of a causal launch IN PHASE would be worth — measured on the shipped library, without a kernel change (developer tool).

profiles/r05_causal_gap.txt leaves 1.8 % (D = 512) ... 4.7 % (D = 1024) of a causal launch to "unequal workgroups".  The suspected mechanism: a head's row tiles
run longest-first on one XCD; the first 32 start together at key 0 and share every L2 fill, each later one starts when an earlier one ends — two KV steps after its
neighbour, at key 0 again — and trails it for life.  If those later tiles walked their keys so that they END together instead (descending from the diagonal; or,
equivalently for the memory system, ascending through a window that is RIGHT-aligned), the stagger of their starts would cancel against the stagger of their lengths:
all of them at the same key at the same time.

The mask-range path can emulate exactly that: a block mask gives row tile i a key window of its causal length — left-aligned [0, len_i) for every tile (arm
"left": the causal launch's memory behaviour) or right-aligned [Nkv - len_i, Nkv) for the tiles of every second round of 32 (arm "phase").  Same kernel build, same
number of workgroups, same KV steps per workgroup, same launch order; only the phase relation of co-resident workgroups differs.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffpa_attn_amd import hip  # noqa: E402
from gpu_causal_gap import timed  # noqa: E402


def windows_mask(N, br, right_aligned_ranks, cus_per_xcd=32):
  nqt = N // br
  tile = torch.arange(N, device="cuda") // br
  length = (tile + 1) * br
  rank = nqt - 1 - tile  # (launch order: longest first)
  right = torch.zeros(N, dtype=torch.bool, device="cuda")
  if right_aligned_ranks:
    right = ((rank // cus_per_xcd) % 2) == 1
  lo = torch.where(right, N - length, torch.zeros_like(length))
  hi = lo + length
  key = torch.arange(N, device="cuda")
  return ((key[None, :] >= lo[:, None]) & (key[None, :] < hi[:, None])).view(1, 1, N, N)


def main():
  B, H, N = 1, 32, 8192
  cus = torch.cuda.get_device_properties(0).multi_processor_count // 8
  for D in (512, 1024, 320):
    torch.manual_seed(0)
    q = torch.randn(B, H, N, D, dtype=torch.bfloat16, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.bfloat16, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.bfloat16, device="cuda")
    plan = {}
    hip.forward(q, k, v, None, True, D ** -0.5, return_lse=False, plan_out=plan)
    br = plan["block_rows"]
    arms, plans = {}, {}
    for name, right in (("left", False), ("phase", True)):
      mask = windows_mask(N, br, right, cus)
      ranges = hip.mask_kv_bounds(mask, N, N)
      plans[name] = {}
      hip.forward(q, k, v, mask, False, D ** -0.5, return_lse=False, kv_bounds=ranges, plan_out=plans[name])
      arms[name] = (lambda m=mask, r=ranges: hip.forward(q, k, v, m, False, D ** -0.5, return_lse=False, kv_bounds=r))
    arms["flag"] = lambda: hip.forward(q, k, v, None, True, D ** -0.5, return_lse=False)
    ts = {a: [] for a in arms}
    for _ in range(3):
      for a, fn in arms.items():
        ts[a].append(timed(fn))
    t = {a: sorted(x)[1] for a, x in ts.items()}
    print(f"PHASE D={D} row tile {br} ({plans['left'].get('kernel')}, tile {plans['left']['block_rows']}x{plans['left']['block_keys']}, splits {plans['left']['splits']}; {cus} CUs per XCD): "
          f"left-aligned windows {t['left']:.1f} us | every second round right-aligned {t['phase']:.1f} us ({(t['left'] / t['phase'] - 1) * 100:+.2f} %) | causal flag {t['flag']:.1f} us "
          f"| all: " + " ".join(f"{a}={'/'.join(f'{x:.0f}' for x in ts[a])}" for a in arms))


if __name__ == "__main__":
  main()
