"""Static rules on the generated gfx950 ISA (needs `python -m ffpa_attn_amd.build --save-temps`, which
__graft_entry__.build() runs; skipped when the assembly files are not there)."""
import glob
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEMPS = glob.glob(os.path.join(ROOT, "ffpa_attn_amd", "csrc", "build", "temps_d*", "*gfx950.s"))


def _tool(name):
  spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


@pytest.mark.skipif(not TEMPS, reason="no --save-temps assembly in csrc/build")
def test_asm_mfma_operands_have_no_valu_writer_inside_the_hazard_window(monkeypatch, capsys):
  """The product kernels issue their S^T MFMAs through inline asm WITHOUT wait-state padding (worth 1.4-2.7 %);
  that is only legal while no VALU instruction writes an A / B / C operand within the two instructions before."""
  monkeypatch.setattr(sys, "argv", ["check_mfma_hazards"])
  assert _tool("check_mfma_hazards").main() == 0, capsys.readouterr().out[-2000:]
  out = capsys.readouterr().out
  assert "asm MFMAs checked" in out and " 0 preceded" in out


@pytest.mark.skipif(not TEMPS, reason="no --save-temps assembly in csrc/build")
def test_headline_kernel_has_no_spill_code_inside_its_mfma_loops(capsys, monkeypatch):
  """D = 512 bf16 prefill kernels: 256 + 256 registers, no scratch and no SGPR lane spills inside the MFMA loops; the build the
  headline workload launches has none between the two loops either."""
  monkeypatch.setattr(sys, "argv", ["isa_stats", "512"])
  _tool("isa_stats").main()
  # template flags after "<D> <ND>": SAFE, DROP, BTILE, mask kind.  The four non-dropout builds of the prefill kernel: without
  # any bias path (mask kind 0: what the headline workload launches), boolean masks only (2), every bias / mask path (1), and
  # bias tiles staged through LDS (BTILE).
  lines = [l for l in capsys.readouterr().out.splitlines() if "bf16  512 1 b0 b0" in l]
  assert len(lines) == 4, lines
  for l in lines:
    assert "vgpr 256 agpr 256" in l and "inside MFMA loops: scratch 0, lane spills 0" in l, l
  headline = [l for l in lines if "512 1 b0 b0 b0 0 " in l]
  assert len(headline) == 1 and "first..last MFMA: scratch ops 0, lane spills 0" in headline[0], headline


@pytest.mark.skipif(not TEMPS, reason="no --save-temps assembly in csrc/build")
def test_16x16x32_builds_have_no_spill_code_inside_their_mfma_loops(capsys, monkeypatch):
  """The builds the headline shape (D = 512, no mask) and config 4 (D = 320, boolean mask) launch: O^T in the AGPRs (D / 2 of
  them), no scratch at all, no SGPR lane spills inside the MFMA loops (one at D = 320); the headline build has none anywhere between its first and
  last MFMA."""
  monkeypatch.setattr(sys, "argv", ["isa_stats", "320", "384", "448", "512"])
  _tool("isa_stats").main()
  lines = [l for l in capsys.readouterr().out.splitlines() if " m16 " in l and " 1 b1 " not in l]  # (the dropout builds carry every bias path and do spill)
  assert len(lines) == 4 * 4, lines  # {bf16, fp16} x {no mask, boolean mask} per head dim
  import re
  for l in lines:
    hot = int(re.search(r"inside MFMA loops: scratch 0, lane spills (\d+)", l).group(1))
    # (D = 320, 128-key tiles: hipcc parks ONE scalar in a VGPR lane inside the QK^T loop of the unmasked build)
    assert "scratch    0 B" in l and hot <= (1 if " 320 " in l else 0), l
  headline = [l for l in lines if "m16 bf16  512 0 " in l]
  assert len(headline) == 1 and "vgpr 256 agpr 256" in headline[0] and "first..last MFMA: scratch ops 0, lane spills 0" in headline[0] and "mfma 256" in headline[0], headline
