"""Host replay of the kernel's LDS addressing (tools/sim_lds_layout.py) for every head dim and tile variant:
the DMA image, the key <-> MFMA-row map pi, the K ds_read_b128 fragments and the V^T ds_read_b64_tr_b16
fragments must pick exactly the documented elements, and every LDS instruction group must be conflict-free
under the bank model of MI355X_MICROARCH.md (the GPU suite confirms SQ_LDS_BANK_CONFLICT = 0 in profiles/)."""
import importlib.util
import os

import pytest

_spec = importlib.util.spec_from_file_location(
    "sim_lds_layout", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "sim_lds_layout.py"))
sim = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(sim)


@pytest.mark.parametrize("D", range(64, 1025, 64))
def test_fragment_maps_and_bank_conflicts(D):
  for d, nd in dict.fromkeys(sim.variants(D)):
    assert sim.check(d, nd) == (1, 1)


@pytest.mark.parametrize("D", range(64, 1025, 64))
def test_fragment_maps_and_bank_conflicts_of_the_16x16x32_build(D):
  assert sim.check_m16(D) == (1, 1)


def test_key_map_is_a_permutation_that_gives_each_lane_half_16_contiguous_keys():
  assert sorted(sim.pi(a) for a in range(32)) == list(range(32))
  for h in range(2):
    for r in range(16):
      a = (r & 3) + 8 * (r >> 2) + 4 * h      # MFMA C-layout row held in register r of lane half h
      assert sim.pi(a) == 16 * h + r
