"""FLOPs model used for every TFLOPS figure this repo reports.

Same definition as the reference's bench (``src/ffpa_attn/cli/_flops.py:15-53``, pinned by
``tests/test_perf_tflops.py:16-55``): forward FLOPs = ``4 * B * Hq * D * valid_pairs`` where a
(query, key) pair is valid iff it is not causally masked, with the causal mask aligned to the KV
tail (``key <= row + Nkv - Nq``).
"""

from __future__ import annotations


def attention_valid_pairs(seqlen_q: int, seqlen_kv: int, causal: bool = False, causal_offset: int | None = None) -> int:
  """Number of (row, key) pairs per (batch, head) that enter the softmax."""
  if seqlen_q <= 0 or seqlen_kv <= 0:
    return 0
  if not causal:
    return seqlen_q * seqlen_kv
  off = seqlen_kv - seqlen_q if causal_offset is None else causal_offset
  # row r sees keys 0 .. min(Nkv - 1, r + off)
  total = 0
  first_full = max(0, seqlen_kv - 1 - off)  # first row that sees every key
  lo_row = max(0, -off)  # first row that sees at least one key
  hi_row = min(seqlen_q, first_full)
  if hi_row > lo_row:
    n = hi_row - lo_row
    a = lo_row + off + 1
    total += n * a + n * (n - 1) // 2
  if seqlen_q > first_full:
    total += (seqlen_q - max(first_full, 0)) * seqlen_kv
  return total


def attention_fwd_flops(
  batch: int, heads_q: int, seqlen_q: int, seqlen_kv: int, head_dim: int, causal: bool = False,
  causal_offset: int | None = None
) -> int:
  """``4 * B * Hq * D * pairs`` (two GEMMs, 2 FLOPs per MAC)."""
  return 4 * batch * heads_q * head_dim * attention_valid_pairs(seqlen_q, seqlen_kv, causal, causal_offset)


def format_tflops(flops: float, seconds: float) -> str:
  if seconds <= 0:
    return "n/a"
  return f"{flops / seconds / 1e12:.1f} TFLOPS"
