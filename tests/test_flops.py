"""FLOPs model parity with the reference's bench (src/ffpa_attn/cli/_flops.py:15-53, pinned there by
tests/test_perf_tflops.py:16-55).  Golden table generated from the reference by make_golden.py."""

import json
import os

from conftest import GOLDEN
from ffpa_attn_amd.flops import attention_fwd_flops, attention_valid_pairs


def test_flops_table_matches_reference():
  rows = json.load(open(os.path.join(GOLDEN, "flops_golden.json")))
  assert len(rows) >= 10
  for r in rows:
    assert attention_valid_pairs(r["Nq"], r["Nkv"], r["causal"]) == r["pairs"], r
    assert attention_fwd_flops(r["B"], r["H"], r["Nq"], r["Nkv"], r["D"], r["causal"]) == r["flops"], r


def test_baseline_config_flops():
  # BASELINE.md §2
  assert attention_fwd_flops(1, 32, 8192, 8192, 512) == 4 * 32 * 512 * 8192 * 8192 == 4398046511104
  assert attention_fwd_flops(1, 32, 8192, 8192, 1024) == 8796093022208
  assert attention_valid_pairs(8192, 2048, True, causal_offset=0) == 14681088  # config 4, SDPA top-left mask
  assert attention_fwd_flops(2, 32, 8192, 2048, 320, True, causal_offset=0) == 4 * 2 * 32 * 320 * 14681088


def test_pairs_brute_force():
  for nq in (1, 3, 8, 17):
    for nkv in (1, 5, 8, 20):
      for off in (None, 0, -3, 4):
        o = nkv - nq if off is None else off
        brute = sum(1 for r in range(nq) for k in range(nkv) if k <= r + o)
        assert attention_valid_pairs(nq, nkv, True, off) == brute
