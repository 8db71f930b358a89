"""ffpa_attn_amd — MI355X-native fused attention forward for large head dims.

Drop-in for the one hot path of xlite-dev/ffpa-attn: ``ffpa_attn_func`` -> Split-D fused
forward.  ``from ffpa_attn_amd import ffpa_attn_func`` replaces ``from ffpa_attn import
ffpa_attn_func``; see INTEGRATION.md.
"""

from .backends import Backend, CUDABackend, CuTeDSLBackend, HIPBackend, SDPABackend, TritonBackend
from .flops import attention_fwd_flops, attention_valid_pairs
from .functional import FFPAAttnMeta
from .interface import ffpa_attn_func, ffpa_attn_varlen_func



def install_alias(force: bool = False) -> bool:
  """Make ``import ffpa_attn`` / ``from ffpa_attn import ffpa_attn_func, TritonBackend`` resolve to THIS package (and ``ffpa_attn.cuda`` to
  the HIP op shim, for call sites that read ``ffpa_attn.cuda.CUDA_FWD_AVAILABLE``) — the one-line switch for a code base written against the
  reference (``src/ffpa_attn/__init__.py:1-14``).  It is a run-time alias in ``sys.modules``, deliberately NOT an ``ffpa_attn/`` directory
  shipped next to this package: a directory would shadow an installed reference for everything on ``sys.path`` behind it, silently.  Returns
  False (and changes nothing) when the reference package is importable, unless ``force``."""
  import importlib.util
  import sys

  from . import hip

  if not force and "ffpa_attn" not in sys.modules and importlib.util.find_spec("ffpa_attn") is not None:
    return False
  if not force and "ffpa_attn" in sys.modules and sys.modules["ffpa_attn"] is not sys.modules[__name__]:
    return False
  sys.modules["ffpa_attn"] = sys.modules[__name__]
  sys.modules["ffpa_attn.cuda"] = hip
  sys.modules[__name__].cuda = hip  # (`import ffpa_attn.cuda as c` reads the submodule as an attribute of the package)
  return True


__version__ = "0.4.0"  # = the library's ffpa_attn_version() ("ffpa-attn-amd 0.4.0 gfx950"; tests/test_capi.py pins the pair)

__all__ = [
  "ffpa_attn_func",
  "ffpa_attn_varlen_func",
  "install_alias",
  "Backend",
  "HIPBackend",
  "CUDABackend",
  "TritonBackend",
  "CuTeDSLBackend",
  "SDPABackend",
  "FFPAAttnMeta",
  "attention_fwd_flops",
  "attention_valid_pairs",
]
