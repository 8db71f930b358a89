"""Per-kernel register / scratch statistics from the --save-temps assembly (python -m ffpa_attn_amd.build --save-temps).

Usage: python tools/isa_stats.py [D ...]     (default: every head dim found)
Prints, per kernel instantiation: VGPRs, AGPRs, SGPRs, scratch bytes, and how many scratch / v_readlane
instructions sit inside the KV-tile loop (between the first and the last MFMA of the kernel).
"""
import glob, os, re, sys

ROOT = os.environ.get("FFPA_ISA_ROOT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ffpa_attn_amd", "csrc", "build")



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


def kernels(path):
  name, body = None, []
  for line in read_isa(path).splitlines(keepends=True):
    m = re.match(r"^(_ZN4ffpa\w+):", line)
    if m:
      if name:
        yield name, body
      name, body = m.group(1), []
    if name:
      body.append(line)
  if name:
    yield name, body


def main():
  dims = sys.argv[1:] or sorted((os.path.basename(d)[7:] for d in glob.glob(os.path.join(ROOT, "temps_d*"))), key=int)
  for d in dims:
    path = isa_files(d)
    if not path:
      continue
    for name, body in (kb for pth in sorted(path) for kb in kernels(pth)):  # (every TU of the head dim: the dense kernels' and the packed-sequence kernel's)
      text = "".join(body)
      get = lambda k: (re.search(rf"; {k}: (\d+)", text) or [None, "?"])[1]
      mf = [i for i, l in enumerate(body) if "v_mfma" in l]
      loop = body[mf[0]:mf[-1]] if mf else []
      n_scr = sum("scratch_" in l for l in loop)
      n_rl = sum("v_readlane" in l or "v_writelane" in l for l in loop)
      # inside the MFMA loops proper: clusters of MFMAs less than 60 lines apart
      hot_rl = hot_scr = 0
      if mf:
        start = prev = mf[0]
        spans = []
        for i in mf[1:]:
          if i - prev > 60:
            spans.append((start, prev))
            start = i
          prev = i
        spans.append((start, prev))
        for a0, b0 in spans:
          hot_rl += sum("v_readlane" in l or "v_writelane" in l for l in body[a0:b0])
          hot_scr += sum("scratch_" in l for l in body[a0:b0])
      short = re.sub(r"_ZN4ffpa23ffpa_fwd_split_d_kernelI(\w+?)EEvNS_7FwdArgsE", r"\1", name)
      short = re.sub(r"_ZN4ffpa19ffpa_fwd_m16_kernelI(\w+?)EEvNS_7FwdArgsE", r"m16 \1", short)
      short = re.sub(r"_ZN4ffpa24ffpa_fwd_m16_pair_kernelI(\w+?)EEvNS_7FwdArgsE", r"m16pair \1", short)
      short = re.sub(r"_ZN4ffpa26ffpa_fwd_m16_varlen_kernelI(\w+?)EEvNS_7FwdArgsENS_10VarlenArgsE", r"m16varlen \1", short)
      short = short.replace("DF16b", "bf16 ").replace("DF16_", "fp16 ").replace("Li", " ").replace("ELb", " b").replace("E", "")
      print(f"D={d:>4} {short:<28} vgpr {get('NumVgprs'):>3} agpr {get('NumAgprs'):>3} sgpr {get('NumSgprs'):>3} "
            f"scratch {get('ScratchSize'):>4} B | first..last MFMA: scratch ops {n_scr}, lane spills {n_rl} | inside MFMA loops: scratch {hot_scr}, lane spills {hot_rl} | mfma {len(mf)}")


if __name__ == "__main__":
  main()
