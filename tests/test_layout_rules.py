"""Repo rules that keep every parity claim honest (checked mechanically):

* the product package never imports / loads anything under oracle/ (no CPU fallback path);
* nothing that runs on the GPU box reads /root/reference;
* the HIP sources contain no CUDA-compat shims, Triton or dual code paths.
"""

import os
import re

from conftest import ROOT

PKG = os.path.join(ROOT, "ffpa_attn_amd")


def _files(root, exts):
  for d, _, fs in os.walk(root):
    if "build" in d.split(os.sep) or "__pycache__" in d:
      continue
    for f in fs:
      if f.endswith(exts):
        yield os.path.join(d, f)


def test_product_never_touches_the_oracle():
  for path in _files(PKG, (".py", ".hip", ".h")):
    text = open(path).read()
    assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), path
    assert "libffpa_oracle" not in text and "ffpa_oracle" not in text, path


def test_gpu_side_never_reads_the_reference_tree():
  runtime = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
  # the two fixture generators run in the authoring container only (their outputs are the committed data under tests/golden/)
  generators = ("make_golden.py", "make_triton_golden.py")
  runtime += [p for p in _files(os.path.join(ROOT, "tests"), (".py",)) if os.path.basename(p) not in generators]
  runtime += list(_files(PKG, (".py",)))
  for path in runtime:
    if path.endswith("test_layout_rules.py") or not os.path.exists(path):
      continue
    text = open(path).read()
    assert "/root/reference" not in text, path


def test_no_compat_layers_in_device_code():
  banned = ("__HIP_PLATFORM_AMD__", "__CUDA_ARCH__", "cuda_runtime", "hipify", "triton", "cutlass", "cute::")
  for path in _files(os.path.join(PKG, "csrc"), (".hip", ".h")):
    text = open(path).read()
    for b in banned:
      if b in ("triton", "cutlass") and b in text.lower():
        # allowed only inside comments that cite the reference
        code = re.sub(r"//.*", "", text)
        assert b not in code.lower(), (path, b)
      elif b not in ("triton", "cutlass"):
        assert b not in text, (path, b)


def test_ops_fail_loudly_without_the_extension(tmp_path, monkeypatch):
  import pytest

  from ffpa_attn_amd import hip

  monkeypatch.setattr(hip, "_lib", None)
  monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "missing.so"))
  with pytest.raises(RuntimeError, match="no fallback"):
    hip.load_library()
