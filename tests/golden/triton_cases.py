"""Inputs of the executed-reference fixtures (tests/golden/ref_triton_cases.npz): the ONE recipe shared by the generator
(make_triton_golden.py, authoring container) and the tests (CPU oracle test, GPU kernel test).  The fixture holds outputs
only; q / k / v / bias are re-created here from numpy's frozen legacy generator."""

import numpy as np

# name, B, Hq, Hkv, Nq, Nkv, D, causal (tail-aligned, the reference's convention), bias shape or None
CASES = [
  ("d320_tail", 1, 2, 2, 200, 333, 320, False, None),
  ("d512_gqa_causal", 1, 2, 1, 192, 320, 512, True, None),
  ("d1024_bias", 1, 1, 1, 130, 257, 1024, False, (1, 1, 130, 257)),
  ("d512_keybias_tail", 2, 2, 2, 72, 777, 512, False, (2, 1, 1, 777)),
]


def triton_case_inputs(case):
  """fp16 q, k, v (and additive fp16 bias) of a case as numpy arrays — the ONE recipe shared by this generator and the tests."""
  name, B, Hq, Hkv, Nq, Nkv, D, causal, bshape = case
  rs = np.random.RandomState(abs(hash_name(name)) % (2 ** 31))
  q = rs.standard_normal((B, Hq, Nq, D)).astype(np.float16)
  k = rs.standard_normal((B, Hkv, Nkv, D)).astype(np.float16)
  v = rs.standard_normal((B, Hkv, Nkv, D)).astype(np.float16)
  bias = None
  if bshape is not None:
    bias = (rs.standard_normal(bshape) * 0.5).astype(np.float16)
    hide = rs.random_sample(bshape) < 0.15  # some -inf entries, never a whole row
    hide[..., 0] = False
    bias = np.where(hide, np.float16(-np.inf), bias).astype(np.float16)
  return q, k, v, bias


def hash_name(name: str) -> int:
  h = 2166136261
  for ch in name.encode():
    h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
  return h
