"""Static rules on the generated gfx950 ISA (needs `python -m ffpa_attn_amd.build --save-temps`, which
__graft_entry__.build() runs; skipped when the assembly files are not there)."""
import glob
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEMPS = glob.glob(os.path.join(ROOT, "ffpa_attn_amd", "csrc", "build", "temps_d*", "*gfx950.s*"))  # (.s, or .s.gz as build() leaves them)


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


def _stats(monkeypatch, capsys, *dims):
  monkeypatch.setattr(sys, "argv", ["isa_stats", *[str(d) for d in dims]])
  _tool("isa_stats").main()
  return capsys.readouterr().out.splitlines()


@pytest.mark.skipif(not TEMPS, reason="no --save-temps assembly in csrc/build")
def test_headline_kernel_has_no_spill_code(capsys, monkeypatch):
  """The kernel the headline workload launches (ffpa_fwd_m16_kernel<bf16, 512, MK=0>): 256 + 256 registers (O^T = the 256 AGPRs), 256
  MFMAs per KV step, no scratch, and not one SGPR lane spill between its first and its last MFMA."""
  lines = _stats(monkeypatch, capsys, 512)
  headline = [l for l in lines if "m16 bf16  512 0 b0" in l]
  assert len(headline) == 1, lines
  assert "vgpr 256 agpr 256" in headline[0] and "scratch    0 B" in headline[0] and "mfma 256" in headline[0], headline
  assert "first..last MFMA: scratch ops 0, lane spills 0" in headline[0], headline
  # head dims >= 128 have ONE prefill family: no 32x32x16 prefill instantiation (ND = 1 at D <= 512) is left in this TU
  assert not [l for l in lines if " m16 " not in l and "  512 1 " in l], lines


@pytest.mark.skipif(not TEMPS, reason="no --save-temps assembly in csrc/build")
def test_prefill_builds_have_no_scratch_inside_their_mfma_loops(capsys, monkeypatch):
  """Every 16x16x32 prefill build (D = 128 ... 1024; no mask / boolean mask / additive bias, with and without dropout): a scratch
  reload inside an MFMA loop would drain the LDS-DMA queue (vmcnt) — none anywhere; the builds without dropout keep their SGPR lane
  spills out of the MFMA loops (a handful at D = 512 in the additive-bias build, whose scalar row tables do not fit)."""
  import re
  every = _stats(monkeypatch, capsys, 128, 192, 256, 320, 384, 448, 512, 640, 1024)
  # the paired-tile kernels (ffpa_fwd_m16_pair_kernel: no bias, without / with dropout): the pass loop around the tile parks a few values in scratch BETWEEN
  # the passes and may reload one in the rare diagonal / ragged-tail branch — never inside an MFMA loop
  paired = [l for l in every if " m16pair " in l]
  assert len(paired) == 9 * 2 * 2, len(paired)
  for l in paired:
    assert "inside MFMA loops: scratch 0" in l and int(re.search(r"first\.\.last MFMA: scratch ops (\d+)", l).group(1)) <= 2, l
  lines = [l for l in every if " m16 " in l]
  assert len(lines) == 9 * 6 * 2, len(lines)  # {MK0, MK2, MK3, MK1, MK0+DROP, MK1+DROP} x {bf16, fp16} per head dim
  for l in lines:
    m = re.search(r"inside MFMA loops: scratch (\d+), lane spills (\d+)", l)
    hot_scratch, hot_lane = int(m.group(1)), int(m.group(2))
    # (the bias + dropout build may reload one value next to the bias conversion at the top of a KV step: the DMA queue is empty there —
    # it sits right behind barrier B's drain — so the reload's vmcnt wait costs nothing; anywhere else it would drain in-flight pieces)
    assert hot_scratch <= (1 if re.search(r" 1 b1 ", l) else 0), l
    drop = bool(re.search(r" [012] b1 ", l))
    mk1 = bool(re.search(r" 1 b[01] ", l))
    if not drop:
      bias_build = mk1 or " 3 b0" in l
      assert hot_lane <= (8 if bias_build else 0), l
      if not bias_build:  # (the bias builds keep a few bytes of scratch for their prologue and their rare ragged-tail / diagonal branch)
        # no scratch access between the first and the last MFMA; the unmasked builds reserve no scratch at all (the boolean-mask builds at
        # D = 256 / 320 reserve a 68-byte frame that no instruction touches: hipcc keeps the slots of SGPR spills it later placed in VGPR lanes)
        # (round 5: the boolean-mask build of the split-D tiles runs the softmax pipeline; it parks the mask's row pointer in scratch in front of the
        # KV loop and reloads it inside the mask-read branch only — steps that read the mask wait for their own global loads there anyway)
        piped_mask = " 2 b0" in l and int(re.match(r"D=\s*(\d+)", l).group(1)) > 512
        n_ops = int(re.search(r"first\.\.last MFMA: scratch ops (\d+)", l).group(1))
        assert n_ops <= (2 if piped_mask else 0), l
        size = int(re.search(r"scratch\s+(\d+) B", l).group(1))
        # (the pipelined split-D builds of round 5 keep 12 bytes for their prologue / epilogue — the row-shared softmax's exchange addresses —: no access
        # between the first and the last MFMA, checked above)
        piped = int(re.match(r"D=\s*(\d+)", l).group(1)) > 512
        assert size == 0 or ((" 2 b0" in l or piped) and size <= 128), l
  # the 32x32x16 prefill kernels that are left (D = 64): no spill code inside their MFMA loops either
  small = [l for l in _stats(monkeypatch, capsys, 64) if "bf16  64 1 b0 b0 b0" in l]
  assert len(small) == 3, small  # mask kinds 0 / 2 / 1
  for l in small:
    assert "inside MFMA loops: scratch 0, lane spills 0" in l, l


@pytest.mark.skipif(not TEMPS, reason="no --save-temps assembly in csrc/build")
def test_packed_sequence_kernels_keep_their_mfma_loops_free_of_spill_code(capsys, monkeypatch):
  """ffpa_fwd_m16_varlen_kernel (one per head dim 128 ... 1024 and dtype; a TU of its own next to the dense one): the dense tile text on per-sequence
  arguments — the sequence's lengths and base pointers live in scalar registers, the register file is as full as the dense kernel's (256 + 256), and the
  one or two values the allocator parks in scratch are written in front of the KV loop and read back behind it: never inside an MFMA loop."""
  import re

  lines = [l for l in _stats(monkeypatch, capsys, 128, 192, 256, 320, 384, 448, 512, 576, 640, 704, 768, 832, 896, 960, 1024) if " m16varlen " in l]
  assert len(lines) == 15 * 2 * 2, len(lines)  # head dims x dtypes x {plain, NT: the non-temporal hint on its K / V pieces}
  for l in lines:
    assert "inside MFMA loops: scratch 0, lane spills 0" in l, l
    assert int(re.search(r"first\.\.last MFMA: scratch ops (\d+)", l).group(1)) == 0, l
    assert int(re.search(r"scratch\s+(\d+) B", l).group(1)) <= 96, l  # (D = 320: 68 B since the row offsets of the packed (head, token) rows joined the prologue)
  d512 = [l for l in lines if "bf16  512" in l]
  assert len(d512) == 2 and all("vgpr 256 agpr 256" in l and "mfma 256" in l for l in d512), d512


@pytest.mark.skipif(not TEMPS, reason="no --save-temps assembly in csrc/build")
def test_the_isa_check_fails_on_a_planted_hazard(tmp_path, capsys):
  """build() keeps only the gzip-compressed device assembly of every TU — and its ISA check must still bite: a copy of one TU's assembly with a VALU
  write to an asm MFMA's operand planted right in front of it, an early reader of an asm MFMA's result, and an M0 write outside the DMA asm are all
  reported (exit status 1); the untouched copy passes."""
  chk = _tool("check_mfma_hazards")
  src = chk.isa_files(64)
  assert src, "D = 64 assembly"
  text = chk.read_isa(src[0])
  lines = text.split("\n")
  # the first inline-asm VGPR-form MFMA of the file: "v_mfma_... v[a:b], v[c:d], v[e:f], ..."
  import re
  idx = next(i for i, l in enumerate(lines) if l.strip().startswith("v_mfma") and re.match(r"\s*v_mfma\S+\s+v\[\d+:\d+\], v\[(\d+):\d+\]", l)
             and any(x.strip().startswith(";;#ASMSTART") for x in lines[max(0, i - 4):i]))
  a_reg = int(re.match(r"\s*v_mfma\S+\s+v\[\d+:\d+\], v\[(\d+):\d+\]", lines[idx]).group(1))
  d_reg = int(re.match(r"\s*v_mfma\S+\s+v\[(\d+):\d+\]", lines[idx]).group(1))

  def run(mutated):
    d = tmp_path / "temps_d64"
    d.mkdir(exist_ok=True)
    (d / "x-gfx950.s").write_text("\n".join(mutated))
    old_root, old_argv = chk.ROOT, sys.argv
    chk.ROOT, sys.argv = str(tmp_path), ["check_mfma_hazards", "64"]
    try:
      rc = chk.main()
    finally:
      chk.ROOT, sys.argv = old_root, old_argv
    return rc, capsys.readouterr().out

  rc, out = run(lines)
  assert rc == 0, out[-1500:]
  rc, out = run(lines[:idx] + [f"\tv_mov_b32_e32 v{a_reg}, 0"] + lines[idx:])
  assert rc == 1 and "HAZARD" in out, out[-1500:]
  rc, out = run(lines[:idx + 1] + [f"\tv_add_f32_e32 v0, v{d_reg}, v{d_reg}"] + lines[idx + 1:])
  assert rc == 1 and "READ-AFTER-MFMA" in out, out[-1500:]
  end = next(i for i in range(idx, len(lines)) if lines[i].strip().startswith(";;#ASMEND"))
  rc, out = run(lines[:end + 1] + ["\ts_mov_b32 m0, 0"] + lines[end + 1:])
  assert rc == 1 and "M0" in out, out[-1500:]
