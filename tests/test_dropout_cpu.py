"""Dropout building blocks on CPU: Philox4x32-10 known answers, the torch-ops mask generator used by the
recompute backward vs the oracle's, and the oracle's dropout statistics."""

import random

import numpy as np
import torch

from ffpa_attn_amd.philox import dropout_keep_mask, philox4x32_10
from oracle import ffpa_oracle as fo


def test_philox_known_answers():
  # Random123 kat_vectors: philox4x32-10, counter 0 / key 0, and the all-ones vector restricted to the
  # (lo, hi, 0, 0) counters this path uses is covered by the cross-check below
  assert fo.philox4x32_10(0, 0) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)


def test_torch_philox_matches_oracle_philox():
  random.seed(3)
  for _ in range(300):
    seed, quad = random.getrandbits(64), random.getrandbits(random.choice([8, 31, 33, 50, 62]))
    want = fo.philox4x32_10(seed, quad)
    got = tuple(int(t.item()) for t in philox4x32_10(seed, torch.tensor([quad], dtype=torch.int64)))
    assert got == want, (seed, quad)


def test_keep_mask_convention_and_rate():
  seed, offset, p = 0x1234_5678_9ABC_DEF0, 44, 0.3
  idx = torch.arange(0, 4096, dtype=torch.int64)
  keep = dropout_keep_mask(seed, offset, idx, p)
  for e in (0, 1, 2, 3, 5, 1023, 4095):  # word (e+offset)&3 of block (e+offset)>>2, u = (w+1)*2^-32 > p
    w = fo.philox4x32_10(seed, (e + offset) >> 2)[(e + offset) & 3]
    assert bool(keep[e]) == bool((np.float32(w) + np.float32(1.0)) * np.float32(2.3283064365386963e-10) > np.float32(p))
  assert abs(keep.float().mean().item() - (1 - p)) < 0.03


def test_oracle_dropout_statistics_and_determinism():
  rng = np.random.default_rng(5)
  B, H, Nq, Nkv, D = 1, 2, 64, 512, 64
  q = fo.to_bits(np.zeros((B, H, Nq, D)), "bf16")            # uniform attention: every P = 1/Nkv
  k = fo.to_bits(rng.standard_normal((B, H, Nkv, D)), "bf16")
  ones = fo.to_bits(np.ones((B, H, Nkv, D)), "bf16")
  _, o32, lse = fo.oracle_forward(q, k, ones, "bf16", dropout_p=0.25, philox_seed=7, philox_offset=8)
  # O = (#kept / Nkv) / (1 - p) per row: mean 1, binomial spread sqrt(p/((1-p) Nkv)) ~ 0.026
  assert abs(o32.mean() - 1.0) < 0.01 and 0.01 < o32[..., 0].std() < 0.05
  np.testing.assert_allclose(lse, np.log(Nkv), atol=1e-5)      # LSE is undropped (prefill.cuh:755-756)
  _, again, _ = fo.oracle_forward(q, k, ones, "bf16", dropout_p=0.25, philox_seed=7, philox_offset=8)
  assert np.array_equal(o32, again)
  _, other, _ = fo.oracle_forward(q, k, ones, "bf16", dropout_p=0.25, philox_seed=8, philox_offset=8)
  assert not np.array_equal(o32, other)
  _, none, _ = fo.oracle_forward(q, k, ones, "bf16", dropout_p=0.0)
  np.testing.assert_allclose(none, 1.0, atol=4e-3)
