"""Philox4x32-10 dropout mask in torch ops — the exact mask the HIP kernel applies.

Used only by the recompute backward (``backward.py``): PyTorch-ROCm's fused attention backward draws
its dropout mask from a different generator convention than the SDPA/cuRAND one the reference and this
kernel use (the reference skips its dropout-parity test on ROCm for that reason,
tests/test_ffpa_fwd.py:339-342), so a training step with ``dropout_p > 0`` has to rebuild the forward's
mask itself.  Convention (csrc/cuffpa/native/prefill.cuh:398-452): element ``e = offset + ((b*Hq + hq)*Nq +
q)*Nkv + k`` uses word ``e & 3`` of the Philox block with counter ``(e >> 2, 0, 0)`` and key ``seed``;
``u = (word + 1) * 2^-32`` in fp32; keep iff ``u > p``.
"""

from __future__ import annotations

import torch

_M32 = 0xFFFFFFFF


def _mulhilo(a: int, x: torch.Tensor):
  """(hi, lo) 32-bit words of a * x for a < 2^32 and x an int64 tensor holding values < 2^32.
  The int64 product wraps mod 2^64, which leaves both words of the true 64-bit product intact."""
  # split x to stay clear of signed-overflow ambiguity: a*x = a*(xh*2^16 + xl)
  xl, xh = x & 0xFFFF, x >> 16
  lo_part = a * xl                      # < 2^48
  hi_part = a * xh                      # < 2^48
  total_lo = lo_part + ((hi_part & 0xFFFF) << 16)   # < 2^49
  lo = total_lo & _M32
  hi = ((hi_part >> 16) + (total_lo >> 32)) & _M32
  return hi, lo


def philox4x32_10(seed: int, quad: torch.Tensor):
  """Four int64 tensors (values < 2^32): the Philox4x32-10 block for counters (quad_lo, quad_hi, 0, 0)."""
  c0, c1 = quad & _M32, (quad >> 32) & _M32
  c2, c3 = torch.zeros_like(quad), torch.zeros_like(quad)
  k0, k1 = seed & _M32, (seed >> 32) & _M32
  for _ in range(10):
    hi0, lo0 = _mulhilo(0xD2511F53, c0)
    hi1, lo1 = _mulhilo(0xCD9E8D57, c2)
    c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
    k0 = (k0 + 0x9E3779B9) & _M32
    k1 = (k1 + 0xBB67AE85) & _M32
  return c0, c1, c2, c3


def dropout_keep_mask(seed: int, offset: int, element_index: torch.Tensor, p: float) -> torch.Tensor:
  """bool tensor: True where the element survives dropout. ``element_index`` = int64 logical score
  indices ``((b*Hq + hq)*Nq + q)*Nkv + k`` (without ``offset``)."""
  e = element_index + offset
  w = philox4x32_10(seed, e >> 2)
  lane = e & 3
  word = torch.where(lane == 0, w[0], torch.where(lane == 1, w[1], torch.where(lane == 2, w[2], w[3])))
  u = (word.to(torch.float32) + 1.0) * 2.3283064365386963e-10
  return u > p
