"""Static check of the inline-asm MFMAs in the generated ISA (python -m ffpa_attn_amd.build --save-temps).

The S^T MFMAs are emitted through inline asm, invisible to hipcc's hazard recognizer.  The kernel relies on
"no VALU instruction writes an operand register of an asm MFMA within the few instructions before it".  This
script parses every asm MFMA of every kernel instantiation and looks back over the preceding WINDOW
instructions (s_waitcnt / s_nop count as instructions, they only help) for a VALU (v_*) instruction whose
destination overlaps the MFMA's A, B or C source registers.  Exit status 1 if any is found.

Usage: python tools/check_mfma_hazards.py [D ...]
"""
import glob, os, re, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ffpa_attn_amd", "csrc", "build")
WINDOW = 2  # the hazard needs 2 wait states; every instruction in between provides at least one
READ_WAIT = 18  # wait states between an MFMA and a non-accumulator reader of its result (the kernels pad s_nop 15 + s_nop 3 = 20)
REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")



def isa_files(d):
  """The device assembly of head dim d's TU: kept gzip-compressed by ffpa_attn_amd.build (10 : 1 — the repo snapshot travels to the GPU box on every
  run), plain while a developer build (tools/dev_compile.sh) is being looked at."""
  base = os.path.join(ROOT, f"temps_d{d}")
  return glob.glob(os.path.join(base, "*gfx950.s")) or glob.glob(os.path.join(base, "*gfx950.s.gz"))


def read_isa(path):
  if path.endswith(".gz"):
    import gzip

    with gzip.open(path, "rt") as f:
      return f.read()
  with open(path) as f:
    return f.read()


def regs(tok):
  m = REG.fullmatch(tok.strip())
  if not m:
    return None
  if m.group(3) is not None:
    return (int(m.group(3)), int(m.group(3)))
  return (int(m.group(1)), int(m.group(2)))


def overlap(a, b):
  return a and b and a[0] <= b[1] and b[0] <= a[1]


def main():
  dims = sys.argv[1:] or sorted((os.path.basename(d)[7:] for d in glob.glob(os.path.join(ROOT, "temps_d*"))), key=int)
  bad = total = 0
  for d in dims:
    for path in isa_files(d):
      lines = read_isa(path).split("\n")
      in_asm = False
      for i, l in enumerate(lines):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
          in_asm = True
        elif t.startswith(";;#ASMEND"):
          in_asm = False
        if not (in_asm and t.startswith("v_mfma")):
          continue
        total += 1
        ops = [x.strip() for x in t.split(None, 1)[1].split(",")]
        srcs = [regs(x) for x in ops[1:4]]
        seen, j = 0, i - 1
        while j >= 0 and seen < WINDOW:
          p = lines[j].strip()
          j -= 1
          if not p or p.startswith((";", ".", "//")) or p.endswith(":"):
            continue
          seen += 1
          if p.startswith("v_") and not p.startswith("v_mfma"):
            dst = regs(p.split(None, 1)[1].split(",")[0]) if len(p.split(None, 1)) > 1 else None
            if any(overlap(dst, s) for s in srcs):
              bad += 1
              print(f"HAZARD D={d} line {i + 1}: `{p}` writes an operand of `{t}`")
  print(f"{total} asm MFMAs checked over D = {', '.join(dims)}: {bad} preceded by a VALU write to an operand")
  # The other direction: the result registers of an asm MFMA must not be READ by anything but a dependent MFMA's accumulator operand until
  # READ_WAIT wait states have passed (MFMA result -> VALU / LDS / VMEM reader; an MFMA taking them as A / B needs them too).  The compiler
  # does not know these registers are MFMA results: a register copy it places for a loop-carried value is enough to break the rule (round 4).
  raw_bad = 0
  for d in dims:
    for path in isa_files(d):
      lines = read_isa(path).split("\n")
      in_asm = False
      for i, l in enumerate(lines):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
          in_asm = True
        elif t.startswith(";;#ASMEND"):
          in_asm = False
        if not (in_asm and t.startswith("v_mfma")):
          continue
        dst = regs(t.split(None, 1)[1].split(",")[0])
        if dst is None:  # AGPR accumulators (O^T): read by the epilogue / the rescale asm only, behind their own pads
          continue
        states, j = 0, i + 1
        while j < len(lines) and states < READ_WAIT:
          p = lines[j].strip()
          j += 1
          if not p or p.startswith((";", ".", "//")) or p.endswith(":"):
            continue
          if p.startswith("s_nop"):
            states += int(p.split()[1]) + 1
            continue
          toks = [x.strip() for x in p.split(None, 1)[1].split(",")] if len(p.split(None, 1)) > 1 else []
          if p.startswith("v_mfma"):
            used = [regs(x.split()[0]) for x in toks[1:3]]  # A and B; the accumulator operand may be the same registers (back-to-back chain)
            states += 4
          else:
            used = [regs(x.split()[0]) if x else None for x in toks]
            states += 1
          if any(overlap(dst, u) for u in used):
            raw_bad += 1
            print(f"READ-AFTER-MFMA D={d} line {j}: `{p}` touches the result of `{t}` (line {i + 1}) after {states - 1} wait states")
            break
  print(f"asm MFMA results read inside {READ_WAIT} wait states: {raw_bad}")
  bad += raw_bad
  # M0 is carried from one LDS-DMA asm statement to the next: nothing outside the asm blocks may write it
  m0_bad = 0
  for d in dims:
    for path in isa_files(d):
      in_asm = False
      for i, l in enumerate(read_isa(path).split("\n")):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
          in_asm = True
        elif t.startswith(";;#ASMEND"):
          in_asm = False
        elif not in_asm and re.match(r"^(s_|v_readfirstlane|v_readlane)\S*\s+m0\b", t):
          m0_bad += 1
          print(f"M0 D={d} line {i + 1}: `{t}` writes m0 outside the LDS-DMA asm")
  print(f"M0 writers outside inline asm: {m0_bad}")
  return 1 if (bad or m0_bad) else 0


if __name__ == "__main__":
  sys.exit(main())
