"""ffpa_attn_amd — MI355X-native fused attention forward for large head dims.

Drop-in for the one hot path of xlite-dev/ffpa-attn: ``ffpa_attn_func`` -> Split-D fused
forward.  ``from ffpa_attn_amd import ffpa_attn_func`` replaces ``from ffpa_attn import
ffpa_attn_func``; see INTEGRATION.md.
"""

from .backends import Backend, CUDABackend, CuTeDSLBackend, HIPBackend, SDPABackend, TritonBackend
from .flops import attention_fwd_flops, attention_valid_pairs
from .functional import FFPAAttnMeta
from .interface import ffpa_attn_func

__version__ = "0.4.0"  # = the library's ffpa_attn_version() ("ffpa-attn-amd 0.4.0 gfx950"; tests/test_capi.py pins the pair)

__all__ = [
  "ffpa_attn_func",
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
