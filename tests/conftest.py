"""Shared fixtures.  `-m "not gpu"` runs on CPU in minutes; `-m gpu` needs one MI355X."""

from __future__ import annotations

import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu`)")


@pytest.fixture(scope="session")
def small_cases():
  """Committed small fixtures: list of dicts with numpy arrays (see tests/golden/make_golden.py)."""
  meta = json.load(open(os.path.join(GOLDEN, "small_cases.json")))
  z = np.load(os.path.join(GOLDEN, "small_cases.npz"))
  cases = []
  for m in meta:
    c = dict(m)
    c["mask_kind"] = c.pop("mask")  # the JSON's "mask" is the kind; the arrays below are the data
    for key in ("q", "k", "v", "o_sdpa", "lse_f64", "mask", "mask_bits"):
      full = f"{m['name']}.{key}"
      if full in z.files:
        c[key] = z[full]
    cases.append(c)
  return cases


def case_bias(c):
  """Additive fp32 bias [B|1,Hq|1,Nq|1,Nkv|1] of a small case (or None) + the torch-side mask."""
  from oracle import ffpa_oracle as fo

  if "mask" in c:
    m = c["mask"]
    if m.dtype == np.bool_:
      return np.where(m, 0.0, -np.inf).astype(np.float32)
    return m.astype(np.float32)
  if "mask_bits" in c:
    return fo.from_bits(c["mask_bits"], c["dtype"]).astype(np.float32)
  return None
