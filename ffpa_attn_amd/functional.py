"""Dispatch state for one ``ffpa_attn_func`` call: kwarg parsing, the SDPA-fallback predicate,
input validation and mask normalisation.

Behavioural mirror of the reference's ``FFPAAttnMeta`` (``src/ffpa_attn/functional.py:593-943``):
same decisions, same exception classes and message fragments (the reference's tests match on
``"unexpected keyword"``, ``"seqlen"``, ``"num_heads"``, ``"is_causal"``, ``"enable_gqa=False"``;
tests/test_ffpa_fwd.py:162-177,1146-1152,1199-1215,1366-1382).  The structure is this repo's own:
a small frozen plan object instead of a mutable meta threaded through an autograd Function.
"""

from __future__ import annotations

import logging
import math
import os
from dataclasses import dataclass, field

import torch

from .backends import Backend, CUDABackend, CuTeDSLBackend, HIPBackend, SDPABackend, coerce_backend

logger = logging.getLogger("FFPA")

SMALL_HEAD_DIM_MAX = 256  # D <= 256 is served by SDPA's own flash kernels (functional.py:68)
SMALL_HEAD_DIM_MIN = 64
MAX_HEAD_DIM = 1024

_warned: set[str] = set()


def warning_once(msg: str) -> None:
  if msg not in _warned:
    _warned.add(msg)
    logger.warning(msg)


def _env_flag(name: str) -> bool:
  return bool(int(os.environ.get(name, "0")))


def _allows_small_d(backend: Backend, head_dim: int) -> bool:
  """Opt-in small-D routing, one env switch per backend name (functional.py:71-105)."""
  if not (SMALL_HEAD_DIM_MIN <= head_dim <= SMALL_HEAD_DIM_MAX):
    return False
  env = {
    "hip": "FFPA_HIP_ALLOW_SMALL_D",
    "cuda": "FFPA_CUDA_ALLOW_SMALL_D",
    "triton": "FFPA_TRITON_ALLOW_SMALL_D",
    "cutedsl": "FFPA_CUTE_ALLOW_SMALL_D",
  }.get(backend.name)
  return bool(env) and (_env_flag(env) or _env_flag("FFPA_HIP_ALLOW_SMALL_D"))


def _allows_short_seq(backend: Backend, Nq: int, Nkv: int, is_causal: bool) -> bool:
  """Opt-in routing of SHORT sequences (8 <= Nq < 512 or Nkv < 512) to the HIP kernel: ``FFPA_HIP_ALLOW_SHORT_SEQ=1`` — the analogue of the small-D switches above for
  the reference's other two length thresholds (functional.py:717-724), which are tuned for GPUs whose SDPA has a flash kernel for such calls; on MI355X SDPA serves a
  large head dim through its math / efficient backends (4 ... 10 x slower than this kernel: chunked prefill of 128 ... 511 rows against a long context).  Off by default
  — the dispatch decisions are the reference's — and never for a call whose answer would change: under ``is_causal`` SDPA masks top-left, this kernel tail-aligned,
  the two agree only for Nq == Nkv."""
  if backend.name not in ("hip", "cuda", "triton") or not _env_flag("FFPA_HIP_ALLOW_SHORT_SEQ"):
    return False
  return (not is_causal) or Nq == Nkv


@dataclass
class AttentionMeta:
  is_causal: bool = False
  dropout_p: float = 0.0
  scale: float | None = None
  is_grad_enabled: bool = False


@dataclass
class FFPAAttnMeta:
  """Non-tensor options of one call (reference: functional.py:593-609)."""

  attn_meta: AttentionMeta = field(default_factory=AttentionMeta)
  forward_meta: Backend = field(default_factory=lambda: HIPBackend(forward=True))
  backward_meta: Backend = field(default_factory=lambda: SDPABackend(backward=True))

  # ------------------------------------------------------------------ construction
  @classmethod
  def from_kwargs(cls, **kwargs) -> "FFPAAttnMeta":
    """Pop ``backend`` / ``forward_backend`` / ``backward_backend``; anything else is a
    ``TypeError`` (functional.py:611-652)."""
    backend = kwargs.pop("backend", None)
    fwd = kwargs.pop("forward_backend", None)
    bwd = kwargs.pop("backward_backend", None)
    fwd = None if fwd is None else coerce_backend(fwd, source="forward_backend")
    bwd = None if bwd is None else coerce_backend(bwd, source="backward_backend")
    if kwargs:
      unexpected = ", ".join(sorted(kwargs))
      raise TypeError(f"ffpa_attn_func() got unexpected keyword argument(s): {unexpected}")
    if fwd is None and bwd is None and backend is not None:
      fwd = bwd = coerce_backend(backend, source="backend")
    if fwd is None:
      fwd = HIPBackend(forward=True)
    if bwd is None:
      bwd = CuTeDSLBackend() if fwd.name == "cutedsl" else SDPABackend(backward=True)
    if not fwd.forward:
      raise AssertionError("forward_backend must be configured with forward=True")
    if (fwd.name == "cutedsl") != (bwd.name == "cutedsl"):
      raise ValueError("forward_backend='cutedsl' requires backward_backend='cutedsl' (and vice versa)")
    return cls(forward_meta=fwd, backward_meta=bwd)

  # ------------------------------------------------------------------ fallback predicate
  def fallback(self, query: torch.Tensor, key: torch.Tensor, attn_mask, dropout_p: float, is_causal: bool = False) -> bool:
    """True when the call must go to ``torch._C._nn.scaled_dot_product_attention``
    (functional.py:676-724).  ``is_causal`` (not an argument of the reference's method) only matters under ``FFPA_HIP_ALLOW_SHORT_SEQ=1``."""
    assert query.dim() == 4, "Expected query shape [B, Nh_q, Nq, D]"
    assert key.dim() == 4, "Expected key shape [B, Nh_kv, Nkv, D]"
    _, _, Nq, D = query.shape
    _, _, Nkv, Dk = key.shape
    assert D == Dk, "Query and key must have the same head dimension"
    fwd = self.forward_meta
    if fwd.name == "sdpa":
      return True
    if fwd.name == "cutedsl":
      return True  # no CuTe-DSL hardware here (reference: cute_forward_available() is False)
    short = (8 <= Nq < 512 or Nkv < 512) and not _allows_short_seq(fwd, Nq, Nkv, is_causal)
    reasons = [
      D <= SMALL_HEAD_DIM_MAX and not _allows_small_d(fwd, D),
      D > MAX_HEAD_DIM,
      short,
    ]
    return any(reasons)  # (large-D CPU tensors reach the op and raise NotImplementedError, like the reference)

  # ------------------------------------------------------------------ validation
  def normalize_inputs(self, query, key, value, attn_mask, dropout_p, is_causal, scale, enable_gqa) -> "FFPAAttnMeta":
    """Validate in the reference's order with its messages (functional.py:726-849)."""
    if not 0.0 <= dropout_p <= 1.0:
      raise ValueError(f"ffpa_attn_func: dropout_p must be in [0, 1], got {dropout_p}")
    if dropout_p >= 1.0:
      raise ValueError("ffpa_attn_func: dropout_p=1.0 is not supported by SDPA fused kernels")
    if attn_mask is not None and is_causal:
      raise RuntimeError("ffpa_attn_func: explicit attn_mask should not be set when is_causal=True")
    if attn_mask is not None and attn_mask.dtype == torch.bool and attn_mask.requires_grad:
      raise TypeError("ffpa_attn_func: boolean attn_mask cannot require gradients")
    # (the reference propagates is_causal to its CUDA backend object and resolves the *_hybrid switches here: functional.py:780-795)
    fwd = self.forward_meta
    if isinstance(fwd, CUDABackend):
      fwd.is_causal = bool(is_causal)
      if fwd.fp8_hybrid is None:
        fwd.fp8_hybrid = bool(fwd.enable_fp8 and is_causal)
      if fwd.fp4_hybrid is None:
        fwd.fp4_hybrid = bool(fwd.enable_fp4 and is_causal)
    if getattr(fwd, "quantized", None):
      # the one pair of switches that cannot be "accepted and ignored": they change the arithmetic (sm_120 FP8 / NVFP4 kernels in the reference)
      raise NotImplementedError(
        f"ffpa_attn_func: enable_{fwd.quantized}=True selects the reference's sm_120 quantised kernels; the MI355X build computes in bf16 / fp16 only"
      )
    if query.dtype not in (torch.float16, torch.bfloat16):
      raise TypeError(f"ffpa_attn_func only supports fp16/bf16, got {query.dtype}")
    if query.dim() != 4 or key.dim() != 4 or value.dim() != 4:
      raise ValueError("query/key/value must be 4-D [B, H, N, D] tensors")
    if query.size(0) != key.size(0) or query.size(0) != value.size(0):
      raise ValueError("query/key/value must share the same batch size")
    if key.size(1) != value.size(1):
      raise ValueError(f"key and value must share the same num_heads, got Nh_k={key.size(1)}, Nh_v={value.size(1)}")
    if query.size(1) % key.size(1) != 0:
      raise ValueError(
        "query num_heads must be an integer multiple of key/value num_heads (GQA/MQA), "
        f"got Nh_q={query.size(1)}, Nh_kv={key.size(1)}"
      )
    if key.size(2) != value.size(2):
      raise ValueError(f"key and value must share the same seqlen, got Nk={key.size(2)}, Nv={value.size(2)}")
    if query.size(3) != key.size(3) or query.size(3) != value.size(3):
      raise ValueError("query/key/value must share the same head dim")
    if not enable_gqa and query.size(1) != key.size(1):
      raise ValueError(
        f"enable_gqa=False but query num_heads ({query.size(1)}) != key/value num_heads ({key.size(1)}). "
        "Set enable_gqa=True or use matching head counts."
      )
    if is_causal and key.size(2) < query.size(2):
      raise ValueError(
        "is_causal=True requires Nkv >= Nq (queries are aligned to the KV tail), "
        f"got Nq={query.size(2)}, Nkv={key.size(2)}"
      )
    self.attn_meta.is_causal = bool(is_causal)
    self.attn_meta.dropout_p = float(dropout_p)
    self.attn_meta.is_grad_enabled = torch.is_grad_enabled()
    self.attn_meta.scale = 1.0 / math.sqrt(query.size(-1)) if scale is None else float(scale)
    return self

  def normalize_attn_mask(self, query, key, attn_mask):
    """SDPA ``attn_mask`` -> compact 4-D bias / mask (functional.py:851-911): validated like the reference; 2-D / 3-D
    masks become broadcasting 4-D views; bool masks stay bool (the kernel applies them)."""
    if attn_mask is None:
      return None
    if attn_mask.device != query.device:
      raise TypeError(
        f"ffpa_attn_func: attn_mask must be on the same device as query, got {attn_mask.device} and {query.device}"
      )
    if attn_mask.dtype not in (torch.bool, torch.float32, query.dtype):
      raise TypeError(
        "ffpa_attn_func: attn_mask dtype must be bool, torch.float32, or match query dtype, "
        f"got attn_mask.dtype={attn_mask.dtype} and query.dtype={query.dtype}"
      )
    B, Hq, Nq, _ = query.shape
    Nkv = key.size(2)
    if attn_mask.dim() not in (2, 3, 4):
      raise ValueError("ffpa_attn_func: attn_mask must be 2-D, 3-D, or 4-D and broadcastable to [B, Nh_q, Nq, Nkv]")
    if attn_mask.size(-2) not in (1, Nq):
      raise ValueError(f"ffpa_attn_func: attn_mask query dimension must be 1 or {Nq}, got {attn_mask.size(-2)}")
    if attn_mask.size(-1) not in (1, Nkv):
      raise ValueError(f"ffpa_attn_func: attn_mask key dimension must be 1 or {Nkv}, got {attn_mask.size(-1)}")
    if attn_mask.dim() >= 3 and attn_mask.size(0) not in (1, B):
      raise ValueError(f"ffpa_attn_func: attn_mask batch dimension must be 1 or {B}, got {attn_mask.size(0)}")
    if attn_mask.dim() == 4 and attn_mask.size(1) not in (1, Hq):
      raise ValueError(f"ffpa_attn_func: 4-D attn_mask head dimension must be 1 or {Hq}, got {attn_mask.size(1)}")
    # Boolean masks reach the kernel as they are (FFPA_BIAS_BOOL8: one byte per score, False = -inf).  The reference
    # materialises a 0 / -inf tensor in query.dtype here (functional.py:891-898) — two extra kernels and a mask-sized
    # allocation per call; the scores are the same.  backward.py builds the additive form only if a gradient needs it.
    bias = attn_mask
    if bias.dim() == 2:
      bias = bias.view(1, 1, bias.size(0), bias.size(1))
    elif bias.dim() == 3:
      bias = bias.view(bias.size(0), 1, bias.size(1), bias.size(2))
    if bias.stride(-1) != 1 and bias.size(-1) != 1:
      bias = bias.contiguous()
    return bias

  def normalize(self, query, key, value, attn_mask, dropout_p, is_causal, scale, enable_gqa):
    self.normalize_inputs(query, key, value, attn_mask, dropout_p, is_causal, scale, enable_gqa)
    return self, query, key, value, self.normalize_attn_mask(query, key, attn_mask)
