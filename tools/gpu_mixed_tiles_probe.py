"""Would a launch that mixes the wide-row (192) and the 128-row tile of D = 320 along the ROW axis pay?  (developer tool; shipped library, no kernel change)

The wide-row tile is ~ 7 % faster per row but its launches come out in other numbers of rounds of workgroups (B1 H32 N8192: 1376 workgroups = 5.4 -> six rounds
against eight of the 128-row tile: - 4 %, so the launch rule keeps the narrow one there).  Two launches back to back — the wide tile on the first rows, the narrow one on
the rest, both sized to WHOLE rounds — would have no ragged round: 32 x 192 + 16 x 128 = 8192 rows: 1024 + 512 workgroups = 4 + 2 rounds.  Priced here with two calls
of the launch wrapper on row slices (what one C-ABI call would do internally).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffpa_attn_amd import hip  # noqa: E402
from gpu_causal_gap import timed  # noqa: E402


def run(tag, B, Hq, Hkv, Nq, Nkv, D, causal, rows_wide):
  torch.manual_seed(0)
  q = torch.randn(B, Hq, Nq, D, dtype=torch.bfloat16, device="cuda")
  k = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  v = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  sc = D ** -0.5
  off = Nkv - Nq if causal == "tail" else 0
  cz = causal is not None
  qa, qb = q[:, :, :rows_wide], q[:, :, rows_wide:]
  plan = {}
  hip.forward(q, k, v, None, cz, sc, causal_offset=off, return_lse=False, plan_out=plan)

  def mixed():
    hip.forward(qa, k, v, None, cz, sc, causal_offset=off, return_lse=False, flags=hip.FLAG_WIDE_TILE, num_splits=1)
    hip.forward(qb, k, v, None, cz, sc, causal_offset=off + rows_wide, return_lse=False, flags=hip.FLAG_NO_WIDE_TILE, num_splits=1)

  def mixed_rev():  # the narrow part first
    hip.forward(qb, k, v, None, cz, sc, causal_offset=off + rows_wide, return_lse=False, flags=hip.FLAG_NO_WIDE_TILE, num_splits=1)
    hip.forward(qa, k, v, None, cz, sc, causal_offset=off, return_lse=False, flags=hip.FLAG_WIDE_TILE, num_splits=1)

  arms = {
    "auto": lambda: hip.forward(q, k, v, None, cz, sc, causal_offset=off, return_lse=False),
    "narrow": lambda: hip.forward(q, k, v, None, cz, sc, causal_offset=off, return_lse=False, flags=hip.FLAG_NO_WIDE_TILE),
    "wide": lambda: hip.forward(q, k, v, None, cz, sc, causal_offset=off, return_lse=False, flags=hip.FLAG_WIDE_TILE),
    "mixed": mixed,
    "mixed_rev": mixed_rev,
  }
  # the two launches of `mixed` must equal the one launch on their rows (same tiles, same order of keys): checked against the forced arms on a few rows
  oa = hip.forward(qa, k, v, None, cz, sc, causal_offset=off, return_lse=False, flags=hip.FLAG_WIDE_TILE, num_splits=1)[0]
  ob = hip.forward(qb, k, v, None, cz, sc, causal_offset=off + rows_wide, return_lse=False, flags=hip.FLAG_NO_WIDE_TILE, num_splits=1)[0]
  ow = hip.forward(q, k, v, None, cz, sc, causal_offset=off, return_lse=False, flags=hip.FLAG_WIDE_TILE, num_splits=1)[0]
  on = hip.forward(q, k, v, None, cz, sc, causal_offset=off, return_lse=False, flags=hip.FLAG_NO_WIDE_TILE, num_splits=1)[0]
  same = bool(torch.equal(torch.nan_to_num(oa), torch.nan_to_num(ow[:, :, :rows_wide]))) and bool(torch.equal(torch.nan_to_num(ob), torch.nan_to_num(on[:, :, rows_wide:])))
  ts = {a: [] for a in arms}
  for _ in range(3):
    for a, fn in arms.items():
      ts[a].append(timed(fn, 80.0))
  t = {a: sorted(x)[1] for a, x in ts.items()}
  best_single = min(t["auto"], t["narrow"], t["wide"])
  print(f"MIXED {tag}: B{B} Hq{Hq} Hkv{Hkv} Nq{Nq} Nkv{Nkv} D{D} causal={causal} | auto plan {plan.get('kernel')} tile {plan['block_rows']}x{plan['block_keys']} splits {plan['splits']} | "
        + " ".join(f"{a} {t[a]:.1f}" for a in arms) + f" us | mixed ({rows_wide} wide rows) vs auto {(t['auto'] / min(t['mixed'], t['mixed_rev']) - 1) * 100:+.2f} %, vs best single {(best_single / min(t['mixed'], t['mixed_rev']) - 1) * 100:+.2f} % | slices bit-equal to the single launches: {same}")


def main():
  cus = torch.cuda.get_device_properties(0).multi_processor_count
  run("self", 1, 32, 32, 8192, 8192, 320, None, 6144)           # 32 x 32 = 1024 wide + 16 x 32 = 512 narrow workgroups: 4 + 2 rounds
  run("self-causal", 1, 32, 32, 8192, 8192, 320, "tail", 6144)
  run("gqa", 1, 32, 8, 8192, 8192, 320, None, 6144)
  run("cfg4-nomask", 2, 32, 8, 8192, 2048, 320, None, 7680)     # 40 x 64 = 2560 wide + 4 x 64 = 256 narrow: 10 + 1 rounds
  run("cfg4-offset0", 2, 32, 8, 8192, 2048, 320, "top", 7680)
  run("N4096", 1, 32, 32, 4096, 4096, 320, None, 3072)          # 16 x 32 = 512 wide + 8 x 32 = 256 narrow: 2 + 1 rounds (narrow alone: 4 rounds)
  run("B2-N4096", 2, 32, 32, 4096, 4096, 320, None, 3072)
  print(f"MIXED ({cus} CUs)")


if __name__ == "__main__":
  main()
