"""In-tree build of the gfx950 C-ABI library (``libffpa_attn_hip.so``).

The reference drives nvcc through setup.py/env.py and generates one TU per
(dtype, acc, headdim, stage) (``env.py:455-521``, ``setup.py:112-145``).  Here the
library has no torch / pybind dependency, so the build is plain ``hipcc``: one
object per head dim (compiled in parallel) + the C-ABI object, linked into a
shared library that stays next to the sources (it travels to the GPU box with the
repo snapshot; a JIT cache would not).

Usage:  python -m ffpa_attn_amd.build [--force] [--jobs N] [--no-test-lib]
"""

from __future__ import annotations

import argparse
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
REPO = os.path.dirname(HERE)
INCLUDE = os.path.join(REPO, "include")
OBJ_DIR = os.path.join(CSRC, "build")
LIB_NAME = "libffpa_attn_hip.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
TEST_LIB_PATH = os.path.join(HERE, "libffpa_attn_hip_test.so")

HEAD_DIMS = [64 * i for i in range(1, 17)]
# head dims whose test-only "safe path" twin kernel (register staging + scalar V gather, used to bisect
# LDS-DMA / transpose-read problems on hardware) is built into libffpa_attn_hip_test.so — never into the product library.
SAFE_HEAD_DIMS = {64, 128, 320, 512, 640, 1024}
# head dims the packed-sequence kernel is built for (FFPA_FOR_EACH_VARLEN_HEAD_DIM, csrc/ffpa_launch.h): the 16x16x32 build's
VARLEN_HEAD_DIMS = [d for d in HEAD_DIMS if d >= 128]

ARCH = "gfx950"
CXXFLAGS = [
  f"--offload-arch={ARCH}",
  "-O3",
  "-std=c++17",
  "-fPIC",
  "-mcode-object-version=5",
  f"-I{INCLUDE}",
  f"-I{CSRC}",
]


def _hipcc() -> str:
  for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError("hipcc not found: the HIP toolchain is required to build ffpa_attn_amd")


def _sources_mtime() -> float:
  paths = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip", ".inc"))]
  paths.append(os.path.join(INCLUDE, "ffpa_attn.h"))
  paths.append(os.path.abspath(__file__))
  return max(os.path.getmtime(p) for p in paths)


def _run(cmd: list[str], cwd: str | None = None) -> None:
  proc = subprocess.run(cmd, capture_output=True, text=True, cwd=cwd)
  if proc.returncode != 0:
    raise RuntimeError("command failed: " + " ".join(cmd) + "\n" + proc.stdout + proc.stderr)


def _check_isa(verbose: bool) -> None:
  """The S^T MFMAs and the LDS-DMA are inline asm (no compiler hazard padding, M0 written behind the compiler's
  back): prove on the ISA just generated that no VALU write lands in an MFMA operand's hazard window and that
  nothing outside the DMA asm touches M0 (tools/check_mfma_hazards.py).  A violation fails the build."""
  import importlib.util

  path = os.path.join(REPO, "tools", "check_mfma_hazards.py")
  if not os.path.exists(path):  # an installed copy without the developer tools
    return
  spec = importlib.util.spec_from_file_location("check_mfma_hazards", path)
  chk = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(chk)
  import contextlib, io

  old_argv, sys.argv = sys.argv, ["check_mfma_hazards"]
  buf = io.StringIO()
  try:
    with contextlib.redirect_stdout(buf):
      rc = chk.main()
  finally:
    sys.argv = old_argv
  if verbose or rc != 0:
    print(buf.getvalue(), end="", flush=True)
  if rc != 0:
    raise RuntimeError("ISA check failed: an inline-asm MFMA operand is written inside its hazard window, or M0 is written outside the DMA asm")


def _prune_temps() -> None:
  """-save-temps leaves ~20 MB per TU (bitcode, preprocessed source, host assembly ...) and the whole tree travels to the GPU box with every
  snapshot.  What the ISA rules (tools/check_mfma_hazards.py, tools/isa_stats.py, tests/test_isa_rules.py) read is the device assembly alone: it
  stays, gzip-compressed (10 : 1); everything else goes.  354 MB -> ~6 MB."""
  import glob
  import gzip

  for tdir in glob.glob(os.path.join(OBJ_DIR, "temps_d*")):
    for path in glob.glob(os.path.join(tdir, "*")):
      if path.endswith("gfx950.s"):
        with open(path, "rb") as src, gzip.open(path + ".gz", "wb", compresslevel=6) as dst:
          shutil.copyfileobj(src, dst)
      elif path.endswith("gfx950.s.gz") and not os.path.exists(path[:-3]):
        continue  # (the assembly of a TU this run did not recompile)
      if os.path.isdir(path):
        shutil.rmtree(path, ignore_errors=True)
      else:
        os.remove(path)


def clean_dev(objects: bool = True) -> None:
  """Remove what only a development session needs and the round-end snapshot should not carry: variant libraries (build_variant) and their objects — and, with
  ``objects``, the object files of the main build once both libraries are linked and fresh (20 MB; ``build()`` is gated on the libraries' mtime, and
  ``build_variant --dims`` rebuilds the objects it finds missing)."""
  import glob

  shutil.rmtree(os.path.join(HERE, "variants"), ignore_errors=True)
  for d in glob.glob(os.path.join(OBJ_DIR, "var_*")):
    shutil.rmtree(d, ignore_errors=True)
  _prune_temps()
  if objects and os.path.exists(LIB_PATH) and os.path.exists(TEST_LIB_PATH):
    newest = _sources_mtime()
    if os.path.getmtime(LIB_PATH) >= newest and os.path.getmtime(TEST_LIB_PATH) >= newest:
      for o in glob.glob(os.path.join(OBJ_DIR, "*.o")):
        os.remove(o)


def build(force: bool = False, jobs: int | None = None, save_temps: bool = True, verbose: bool = True, test_lib: bool = True) -> str:
  """Compile every kernel for gfx950, link ``libffpa_attn_hip.so`` (product kernels only) and — with ``test_lib`` —
  ``libffpa_attn_hip_test.so`` (the same library plus the register-staged SAFE twins behind FFPA_FLAG_DEBUG_SAFE_PATH,
  used by the GPU tests to bisect the LDS-DMA / transpose-read data path); returns the product library's path.
  The generated ISA is kept (``-save-temps``) and checked for the hazards the inline asm hides from the compiler."""
  newest = _sources_mtime()
  have = os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= newest
  have_test = not test_lib or (os.path.exists(TEST_LIB_PATH) and os.path.getmtime(TEST_LIB_PATH) >= newest)
  if not force and have and have_test:
    return LIB_PATH
  hipcc = _hipcc()
  os.makedirs(OBJ_DIR, exist_ok=True)
  jobs = jobs or max(1, (os.cpu_count() or 4))
  tasks: list[tuple[str, list[str], str | None]] = []
  objs: list[str] = []
  test_objs: list[str] = []
  extra = ["-save-temps"] if save_temps else []  # temps land in the compile's cwd (one dir per TU)

  def stale(obj: str) -> bool:
    return force or not os.path.exists(obj) or os.path.getmtime(obj) < newest

  # the product define: the kernel headers then refuse any developer switch that is not at its shipped default
  product = ["-DFFPA_PRODUCT_BUILD=1"]
  for d in HEAD_DIMS:
    obj = os.path.join(OBJ_DIR, f"ffpa_fwd_d{d}.o")
    objs.append(obj)
    if stale(obj):
      tmp = os.path.join(OBJ_DIR, f"temps_d{d}")
      os.makedirs(tmp, exist_ok=True)
      tasks.append((obj, [hipcc, *CXXFLAGS, *product, *extra, f"-DFFPA_INST_D={d}", "-c", os.path.join(CSRC, "ffpa_fwd_inst.hip"), "-o", obj], tmp))
    if test_lib and d in SAFE_HEAD_DIMS:
      tobj = os.path.join(OBJ_DIR, f"ffpa_fwd_d{d}_test.o")
      test_objs.append(tobj)
      if stale(tobj):
        tasks.append((tobj, [hipcc, *CXXFLAGS, *product, f"-DFFPA_INST_D={d}", "-DFFPA_INST_SAFE=1", "-c", os.path.join(CSRC, "ffpa_fwd_inst.hip"), "-o", tobj], None))
    elif test_lib:
      test_objs.append(obj)
  # the packed-sequence kernel (ffpa_varlen_inst.hip): a TU of its own per head dim of the 16x16x32 build, so that the dense kernels' objects stay what they were;
  # its device assembly lands next to the dense TU's (temps_d<D>: the ISA rules read every *.s of a head dim)
  for d in VARLEN_HEAD_DIMS:
    obj = os.path.join(OBJ_DIR, f"ffpa_varlen_d{d}.o")
    objs.append(obj)
    if test_lib:
      test_objs.append(obj)
    if stale(obj):
      tmp = os.path.join(OBJ_DIR, f"temps_d{d}")
      os.makedirs(tmp, exist_ok=True)
      tasks.append((obj, [hipcc, *CXXFLAGS, *product, *extra, f"-DFFPA_INST_D={d}", "-c", os.path.join(CSRC, "ffpa_varlen_inst.hip"), "-o", obj], tmp))
  capi = os.path.join(OBJ_DIR, "ffpa_capi.o")
  objs.append(capi)
  if stale(capi):
    tasks.append((capi, [hipcc, *CXXFLAGS, *product, "-c", os.path.join(CSRC, "ffpa_capi.hip"), "-o", capi], None))
  if test_lib:
    tcapi = os.path.join(OBJ_DIR, "ffpa_capi_test.o")
    test_objs.append(tcapi)
    if stale(tcapi):
      tasks.append((tcapi, [hipcc, *CXXFLAGS, *product, "-DFFPA_INST_SAFE=1", "-c", os.path.join(CSRC, "ffpa_capi.hip"), "-o", tcapi], None))
  if verbose:
    print(f"[ffpa_attn_amd.build] compiling {len(tasks)} objects for {ARCH} with {jobs} jobs", flush=True)
  # longest jobs first (head dims >= 320 carry the 16x16x32 builds as well: 2 - 3 x the compile time of the small ones)
  def cost(t):
    m = re.search(r"ffpa_fwd_d(\d+)", t[0])
    return -(int(m.group(1)) + (2048 if m and int(m.group(1)) >= 320 else 0)) if m else 0

  tasks.sort(key=cost)
  with ThreadPoolExecutor(max_workers=jobs) as pool:
    list(pool.map(lambda t: _run(t[1], cwd=t[2]), tasks))
  if save_temps:
    _check_isa(verbose)
    _prune_temps()
  _run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH, *objs, "-Wl,-rpath,/opt/rocm/lib"])
  if test_lib:
    _run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", TEST_LIB_PATH, *test_objs, "-Wl,-rpath,/opt/rocm/lib"])
  if verbose:
    print(f"[ffpa_attn_amd.build] linked {LIB_PATH}" + (f" and {TEST_LIB_PATH}" if test_lib else ""), flush=True)
  return LIB_PATH


def build_variant(tag: str, defs: list[str], jobs: int | None = None, head_dims: list[int] | None = None) -> str:
  """Developer tool: build ``variants/libffpa_attn_hip_<tag>.so`` with extra ``-D`` tunables (see the
  FFPA_* macros at the top of csrc/ffpa_fwd_kernel.h) for A/B timing with tools/gpu_ab.py.  With
  ``head_dims`` only those head dims are recompiled with the tunables; the rest of the C-ABI table
  links the objects of the main build (which must exist), so a variant costs seconds."""
  hipcc = _hipcc()
  odir = os.path.join(OBJ_DIR, f"var_{tag}")
  os.makedirs(odir, exist_ok=True)
  vdir = os.path.join(HERE, "variants")
  os.makedirs(vdir, exist_ok=True)
  lib = os.path.join(vdir, f"libffpa_attn_hip_{tag}.so")
  newest = _sources_mtime()
  stamp = os.path.join(odir, "defs.txt")
  key = " ".join(defs) + " | " + ",".join(str(d) for d in (head_dims or HEAD_DIMS))
  same_defs = os.path.exists(stamp) and open(stamp).read() == key
  if same_defs and os.path.exists(lib) and os.path.getmtime(lib) >= newest:
    return lib
  if head_dims:
    # the untouched head dims come from the main build (whose objects a round-end clean_dev() may have removed: rebuilt then)
    missing = [d for d in HEAD_DIMS if d not in head_dims and not os.path.exists(os.path.join(OBJ_DIR, f"ffpa_fwd_d{d}.o"))]
    missing += [d for d in VARLEN_HEAD_DIMS if not os.path.exists(os.path.join(OBJ_DIR, f"ffpa_varlen_d{d}.o"))]
    build(force=bool(missing), verbose=False)
  elif any(not os.path.exists(os.path.join(OBJ_DIR, f"ffpa_varlen_d{d}.o")) for d in VARLEN_HEAD_DIMS):
    build(force=True, verbose=False)
  tasks, objs = [], []
  for d in HEAD_DIMS:
    if head_dims and d not in head_dims:
      objs.append(os.path.join(OBJ_DIR, f"ffpa_fwd_d{d}.o"))
      continue
    obj = os.path.join(odir, f"ffpa_fwd_d{d}.o")
    objs.append(obj)
    tasks.append([hipcc, *CXXFLAGS, *defs, f"-DFFPA_INST_D={d}", "-c", os.path.join(CSRC, "ffpa_fwd_inst.hip"), "-o", obj])
  objs += [os.path.join(OBJ_DIR, f"ffpa_varlen_d{d}.o") for d in VARLEN_HEAD_DIMS]  # (the packed-sequence kernels: the main build's, never a variant's)
  capi = os.path.join(odir, "ffpa_capi.o")
  objs.append(capi)
  tasks.append([hipcc, *CXXFLAGS, *defs, "-c", os.path.join(CSRC, "ffpa_capi.hip"), "-o", capi])  # (the plan must see the same tunables as the kernels)
  with ThreadPoolExecutor(max_workers=jobs or (os.cpu_count() or 4)) as pool:
    list(pool.map(_run, tasks))
  _run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib, *objs, "-Wl,-rpath,/opt/rocm/lib"])
  with open(stamp, "w") as f:
    f.write(key)
  return lib


def main() -> None:
  ap = argparse.ArgumentParser(description=__doc__)
  ap.add_argument("--force", action="store_true")
  ap.add_argument("--jobs", type=int, default=None)
  ap.add_argument("--save-temps", action="store_true", help="(default now: the generated ISA is always kept and checked)")
  ap.add_argument("--no-test-lib", action="store_true", help="skip libffpa_attn_hip_test.so (the SAFE twin kernels the GPU tests use)")
  ap.add_argument("--variant", nargs="+", metavar=("TAG", "DEF"), help="build variants/libffpa_attn_hip_TAG.so with -D defs")
  ap.add_argument("--dims", type=lambda s: [int(x) for x in s.split(",")], default=None, help="variant: only recompile these head dims")
  ap.add_argument("--clean-dev", action="store_true", help="remove variants/ and their objects, prune the ISA temps (what a round-end snapshot should not carry)")
  args = ap.parse_args()
  if args.clean_dev:
    clean_dev()
    return
  if args.variant:
    print(build_variant(args.variant[0], [d if d.startswith("-D") else "-D" + d for d in args.variant[1:]], jobs=args.jobs, head_dims=args.dims))
    return
  print(build(force=args.force, jobs=args.jobs, test_lib=not args.no_test_lib))


if __name__ == "__main__":
  sys.exit(main())
