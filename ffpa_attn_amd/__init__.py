"""ffpa_attn_amd — MI355X-native fused attention forward for large head dims.

Drop-in for the one hot path of xlite-dev/ffpa-attn: ``ffpa_attn_func`` -> Split-D fused
forward.  ``from ffpa_attn_amd import ffpa_attn_func`` replaces ``from ffpa_attn import
ffpa_attn_func``; see INTEGRATION.md.
"""

from .backends import Backend, CUDABackend, CuTeDSLBackend, HIPBackend, SDPABackend, TritonBackend
from .flops import attention_fwd_flops, attention_valid_pairs
from .functional import FFPAAttnMeta
from .decode import DecodeStep
from .interface import ffpa_attn_func, ffpa_attn_varlen_func



# reference submodule names that differ from this package's (src/ffpa_attn/cuda/__init__.py, src/ffpa_attn/ffpa_attn_interface.py)
_ALIAS_RENAMES = {"cuda": "hip", "ffpa_attn_interface": "interface"}


class _AliasFinder:
  """Meta-path finder behind install_alias(): ``import ffpa_attn.X`` / ``from ffpa_attn.X import ...`` returns the ALREADY-imported
  ``ffpa_attn_amd.X`` module object (``cuda`` -> ``hip``, ``ffpa_attn_interface`` -> ``interface``) instead of letting the path finder execute
  the file a second time under the other name — a second ``backends`` would define a second ``Backend`` class (isinstance checks in
  FFPAAttnMeta.from_kwargs then fail), a second ``hip`` would define the torch.library op twice."""

  prefix = "ffpa_attn."

  def find_spec(self, fullname, path=None, target=None):
    import importlib
    import importlib.util
    import sys

    if not fullname.startswith(self.prefix) or sys.modules.get("ffpa_attn") is not sys.modules.get(__name__):
      return None
    tail = fullname[len(self.prefix):].split(".")
    tail[0] = _ALIAS_RENAMES.get(tail[0], tail[0])
    real = __name__ + "." + ".".join(tail)
    try:
      mod = importlib.import_module(real)
    except ImportError:
      return None

    # (the import machinery stamps __spec__ / __loader__ / __name__ of the alias on the module it is handed: the loader puts the real ones back)
    real_spec, real_loader = getattr(mod, "__spec__", None), getattr(mod, "__loader__", None)

    class _Existing:
      def create_module(self, spec):
        return mod

      def exec_module(self, module):  # (already executed under its real name)
        module.__spec__, module.__loader__, module.__name__ = real_spec, real_loader, real

    return importlib.util.spec_from_loader(fullname, _Existing(), origin=getattr(mod, "__file__", None))


def install_alias(force: bool = False) -> bool:
  """Make ``import ffpa_attn`` / ``from ffpa_attn import ffpa_attn_func, TritonBackend`` resolve to THIS package, ``ffpa_attn.cuda`` to
  the HIP op shim (for call sites that read ``ffpa_attn.cuda.CUDA_FWD_AVAILABLE``), and every ``ffpa_attn.<submodule>`` (``functional``,
  ``backends``, ``ffpa_attn_interface`` ...) to the SAME module object as ``ffpa_attn_amd.<submodule>`` — the one-line switch for a code base
  written against the reference (``src/ffpa_attn/__init__.py:1-14``).  It is a run-time alias in ``sys.modules`` + a meta-path finder,
  deliberately NOT an ``ffpa_attn/`` directory shipped next to this package: a directory would shadow an installed reference for
  everything on ``sys.path`` behind it, silently.  Returns False (and changes nothing) when the reference package is importable, unless ``force``."""
  import importlib.util
  import sys

  from . import backends, functional, hip, interface  # noqa: F401  (imported once, under their real names, before any alias can reach them)

  if not force and "ffpa_attn" not in sys.modules and importlib.util.find_spec("ffpa_attn") is not None:
    return False
  if not force and "ffpa_attn" in sys.modules and sys.modules["ffpa_attn"] is not sys.modules[__name__]:
    return False
  sys.modules["ffpa_attn"] = sys.modules[__name__]
  # every submodule imported so far, under the reference's name for it as well
  inverse = {v: k for k, v in _ALIAS_RENAMES.items()}
  for name, mod in list(sys.modules.items()):
    if mod is None or not name.startswith(__name__ + "."):
      continue
    tail = name[len(__name__) + 1:].split(".")
    sys.modules["ffpa_attn." + ".".join(tail)] = mod
    if tail[0] in inverse:
      sys.modules["ffpa_attn." + ".".join([inverse[tail[0]]] + tail[1:])] = mod
  sys.modules[__name__].cuda = hip  # (`import ffpa_attn.cuda as c` reads the submodule as an attribute of the package)
  sys.modules[__name__].ffpa_attn_interface = interface
  if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())  # submodules imported later (ffpa_attn.sharding, ffpa_attn.flops ...)
  return True


__version__ = "0.5.0"  # = the library's ffpa_attn_version() ("ffpa-attn-amd 0.5.0 gfx950"; tests/test_capi.py pins the pair)

__all__ = [
  "ffpa_attn_func",
  "ffpa_attn_varlen_func",
  "install_alias",
  "DecodeStep",
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
