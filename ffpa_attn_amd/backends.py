"""Per-call backend configuration objects (the ``backend=`` / ``forward_backend=`` /
``backward_backend=`` kwargs of :func:`ffpa_attn_amd.ffpa_attn_func`).

Mirrors the reference's user-facing config surface (``src/ffpa_attn/functional.py:176-507``:
``Backend``, ``SDPABackend``, ``CUDABackend``, ``TritonBackend``, ``CuTeDSLBackend`` and the
string coercion rules) so that call sites written against the reference keep working: every
field of the reference's dataclasses is accepted under its name, with its default, and checked
by the reference's own assertions (same message) — ``TritonBackend(autotune_mode="max",
enable_tma=True)`` and ``CUDABackend(forward=True, fp8_smooth_k=False)`` construct here exactly as
they do there (pinned against the imported reference: tests/golden/backend_golden.json).  On
MI355X there is one native forward implementation — the hand-written gfx950 kernel — so every
non-SDPA backend name resolves to it (``HIPBackend``); the NVIDIA-only knobs (TMA, CuTe, warp
specialisation, Triton autotune, the backward-only Triton storage options, the FP8 / FP4
quantisation sub-options) select nothing here and are ignored.  The two switches that would change
the ARITHMETIC if they were silently ignored — ``enable_fp8`` / ``enable_fp4`` — construct (as in the
reference) and are refused when a call reaches the kernel with them set.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import torch


_GRAD_STORAGE = {None: None, "fp16": torch.float16, "fp32": torch.float32, torch.float16: torch.float16, torch.float32: torch.float32}


def _normalize_grad_storage_dtype(dtype):
  """``None`` / ``"fp16"`` / ``"fp32"`` / ``torch.float16`` / ``torch.float32`` (functional.py:158-172), the reference's error text otherwise."""
  try:
    return _GRAD_STORAGE[dtype]
  except (KeyError, TypeError):
    raise ValueError(f"grad_kv_storage_dtype must be None, 'fp16', 'fp32', torch.float16, or torch.float32, got {dtype!r}") from None


@dataclass
class Backend:
  """Base config.  ``forward`` / ``backward`` say which pass the instance configures;
  leaving both ``None`` means "both" (reference: functional.py:176-197)."""

  name: str
  forward: bool | None = None
  backward: bool | None = None

  def __post_init__(self) -> None:
    fwd, bwd = self.forward, self.backward
    # a side left open is the complement of the side that was given; both open = both passes
    self.forward = (bwd is None or not bwd) if fwd is None else fwd
    self.backward = (fwd is None or not fwd) if bwd is None else bwd


@dataclass
class SDPABackend(Backend):
  """PyTorch ``scaled_dot_product_attention``.  As a forward backend it always
  short-circuits to the native op (functional.py:200-216, :694-695)."""

  name: str = "sdpa"
  high_precision_grad: bool = False


@dataclass
class HIPBackend(Backend):
  """The MI355X-native forward (``ffpa_attn::_fwd_hip``).

  :ivar acc: accumulator precision; only ``"f32"`` exists on this path (the reference rejects
      bf16 + ``acc="f16"`` too: functional.py:798-803).
  :ivar stages: accepted for compatibility; the LDS pipeline depth is fixed per head dim.
  :ivar rescale_threshold: lazy-rescale threshold in log2 units; ``None`` = the reference's
      ``FFPA_RESCALE_THRESHOLD`` = 8 (csrc/cuffpa/common.cuh:14); ``0`` = exact recurrence.
  :ivar kv_bounds: precomputed key ranges of THIS call's ``attn_mask`` (``ffpa_attn_amd.hip.mask_kv_bounds(mask, Nq, Nkv)``: int32
      ``[B|1, Hq|1, ceil(Nq / 32), 4]``) — the kernel skips the KV tiles the mask hides and does not read the mask where it is neutral.  ``None``
      (default): the ranges are derived from the mask by a scan kernel on EVERY call that is worth one (a full read of the mask per layer and
      step); a serving loop with a static mask scans once and passes the result here — or opts into the version-counter cache with
      ``FFPA_HIP_MASK_BOUNDS_CACHE=1`` (see ``hip.forward``).  The ranges must describe the mask they are passed with: stale ranges are silently
      skipped keys.

  The fields of this class are keyword-only, so that the positional order of ``CUDABackend`` / ``TritonBackend`` stays the reference's.
  """

  name: str = "hip"
  acc: str = field(default="f32", kw_only=True)
  stages: int | None = field(default=None, kw_only=True)
  rescale_threshold: float | None = field(default=None, kw_only=True)
  kv_bounds: "torch.Tensor | None" = field(default=None, kw_only=True, repr=False, compare=False)

  def __post_init__(self) -> None:
    super().__post_init__()
    self._check_acc()

  def _check_acc(self) -> None:
    assert self.acc in ("f16", "f32"), f"acc must be 'f16' or 'f32', got {self.acc!r}"
    if self.acc == "f16":
      # (the reference raises the same class when its fp16-acc kernels were not compiled: functional.py:274-278)
      raise ValueError(
        f"{type(self).__name__}(acc='f16') requires the fp16 MMA acc kernels, which do not exist in the gfx950 build "
        "(the kernel accumulates in fp32 only)."
      )

  @property
  def acc_code(self) -> int:
    return 1

  @property
  def quantized(self) -> str | None:
    """``"fp8"`` / ``"fp4"`` when the instance asks for a quantised forward (refused at dispatch), else ``None``."""
    return None


_CUDA_CHOICES = (
  ("fp8_q_quant_method", ("per_block", "per_thread"), "'per_block' or 'per_thread'"),
  ("fp8_k_quant_method", ("per_block", "per_thread"), "'per_block' or 'per_thread'"),
  ("fp8_v_quant_method", ("per_block", "per_channel"), "'per_block' or 'per_channel'"),
  ("fp8_pv_acc_type", ("f16", "f32"), "'f32' or 'f16'"),
  ("fp8_qk_mm_type", ("fp8", "int8"), "'fp8' or 'int8'"),
)


@dataclass
class CUDABackend(HIPBackend):
  """Source-compatible with the reference's hand-written native backend (functional.py:218-373): the same fields, defaults
  and assertions.  Runs the gfx950 kernel; ``enable_tma`` / ``enable_cute`` / ``enable_ws`` and the ``fp8_*`` / ``fp4_*``
  sub-options select nothing here."""

  name: str = "cuda"
  acc: str = "f32"          # (positional, in the reference's order: name, forward, backward, acc, stages, enable_tma, ...)
  stages: int | None = None
  enable_tma: bool | None = None
  enable_cute: bool | None = None
  enable_ws: bool = False
  enable_fp8: bool = False
  enable_fp4: bool = False
  fp8_smooth_k: bool = True
  fp8_smooth_v: bool = False
  fp8_q_quant_method: str = "per_block"
  fp8_k_quant_method: str = "per_block"
  fp8_v_quant_method: str = "per_block"
  fp8_pv_acc_type: str = "f32"
  fp8_qk_mm_type: str = "fp8"
  fp8_hybrid: bool | None = None
  fp8_hybrid_n_early: int = 256
  fp4_hybrid: bool | None = None
  fp4_hybrid_n_early: int = 256
  is_causal: bool = False  # runtime: set from ffpa_attn_func(is_causal=...) by normalize_inputs, as in the reference

  def __post_init__(self) -> None:
    Backend.__post_init__(self)
    # the reference's CUDA backend is forward-only: CUDABackend(), backend="cuda" and backward_backend="cuda" trip this very assertion
    # (functional.py:266-268 after Backend.__post_init__ :189-196); callers spell it forward_backend="cuda" / CUDABackend(forward=True)
    assert not self.backward, "cuda backend does not support backward"
    self._check_acc()
    assert not (self.enable_fp8 and self.enable_fp4), ("enable_fp8 and enable_fp4 are mutually exclusive")
    for name, allowed, wording in _CUDA_CHOICES:  # (the reference's messages, functional.py:270-300: call sites and tests match on them)
      value = getattr(self, name)
      assert value in allowed, f"{name} must be {wording}, got {value!r}"
    assert self.fp8_v_quant_method == "per_channel" or not self.fp8_smooth_v, ("fp8_smooth_v requires fp8_v_quant_method='per_channel'")
    # the reference resolves enable_tma / enable_cute = None to what its build and device offer (functional.py:304-325): neither exists here
    self.enable_tma, self.enable_cute = bool(self.enable_tma), bool(self.enable_cute)

  @property
  def quantized(self) -> str | None:
    return "fp4" if self.enable_fp4 else ("fp8" if self.enable_fp8 else None)


@dataclass
class TritonBackend(HIPBackend):
  """Source-compatible with the reference's default backend (functional.py:377-421): the same fields, defaults and assertions.
  No Triton is used here; a forward call runs the gfx950 kernel, the backward-only options configure a backward this build
  routes to SDPA's (ffpa_attn_amd/backward.py) and are ignored."""

  name: str = "triton"
  autotune: bool = False
  autotune_mode: str = "fast"
  enable_tma: bool = False
  enable_ws: bool = False
  persist_dkdv: bool = False
  split_launch: bool = False
  preprocess_d_chunk: bool = False
  grad_kv_storage_dtype: torch.dtype | str | None = None
  grad_q_storage_dtype: torch.dtype | str | None = None

  def __post_init__(self) -> None:
    super().__post_init__()
    assert self.autotune_mode in ("fast", "max"), f"Unsupported autotune_mode={self.autotune_mode!r}; choose 'fast' or 'max'."
    self.grad_kv_storage_dtype = _normalize_grad_storage_dtype(self.grad_kv_storage_dtype)
    self.grad_q_storage_dtype = _normalize_grad_storage_dtype(self.grad_q_storage_dtype)
    if self.persist_dkdv:
      assert self.backward, "persist_dkdv is only valid for Triton backward"
      assert self.enable_tma, "persist_dkdv requires enable_tma=True"
    backward_only = (self.split_launch, self.preprocess_d_chunk, self.grad_kv_storage_dtype is not None, self.grad_q_storage_dtype is not None)
    assert self.backward or not any(backward_only), "backward-only Triton options require backward=True"


@dataclass
class CuTeDSLBackend(Backend):
  """NVIDIA CuTe-DSL backend (functional.py:424-470).  Never available on AMD hardware, so —
  exactly like the reference when ``cute_forward_available()`` is false — it falls back to SDPA."""

  name: str = "cutedsl"
  grad_kv_storage_dtype: torch.dtype | str | None = None

  def __post_init__(self) -> None:
    super().__post_init__()
    self.grad_kv_storage_dtype = _normalize_grad_storage_dtype(self.grad_kv_storage_dtype)
    assert self.backward or self.grad_kv_storage_dtype is None, "grad_kv_storage_dtype is a backward-only option; requires backward=True"


_BACKEND_BY_NAME = {
  "hip": HIPBackend,
  "cuda": CUDABackend,
  "triton": TritonBackend,
  "cutedsl": CuTeDSLBackend,
  "sdpa": SDPABackend,
}


def coerce_backend(backend: "Backend | str", *, source: str) -> Backend:
  """``str`` / ``Backend`` -> ``Backend`` with the reference's errors (functional.py:486-507)."""
  if isinstance(backend, str):
    cls = _BACKEND_BY_NAME.get(backend)
    if cls is None:
      raise ValueError(
        f"ffpa_attn_func: {source} must be 'cuda', 'triton', 'cutedsl', or 'sdpa' (or 'hip'), got {backend!r}"
      )
    if source == "backend":
      return cls()
    is_forward = source.startswith("forward")
    return cls(forward=is_forward, backward=not is_forward)
  if not isinstance(backend, Backend):
    raise TypeError(f"ffpa_attn_func: {source} must be a str or Backend instance, got {type(backend).__name__}")
  return backend
