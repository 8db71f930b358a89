"""Per-call backend configuration objects (the ``backend=`` / ``forward_backend=`` /
``backward_backend=`` kwargs of :func:`ffpa_attn_amd.ffpa_attn_func`).

Mirrors the reference's user-facing config surface (``src/ffpa_attn/functional.py:176-507``:
``Backend``, ``SDPABackend``, ``CUDABackend``, ``TritonBackend``, ``CuTeDSLBackend`` and the
string coercion rules) so that call sites written against the reference keep working.  On
MI355X there is exactly one native forward implementation — the hand-written gfx950 kernel —
so every non-SDPA backend name resolves to it (``HIPBackend``); NVIDIA-only knobs (TMA,
CuTe, FP8/FP4, Triton autotune) are accepted for source compatibility and ignored.
"""

from __future__ import annotations

from dataclasses import dataclass


@dataclass
class Backend:
  """Base config.  ``forward`` / ``backward`` say which pass the instance configures;
  leaving both ``None`` means "both" (reference: functional.py:176-197)."""

  name: str
  forward: bool | None = None
  backward: bool | None = None

  def __post_init__(self) -> None:
    if self.forward is None and self.backward is None:
      self.forward = self.backward = True
    elif self.forward is None:
      self.forward = not self.backward
    elif self.backward is None:
      self.backward = not self.forward


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
  :ivar causal_offset: ``None`` = tail-aligned causal mask (reference semantics,
      ``key <= row + Nkv - Nq``); ``0`` = SDPA's top-left alignment (only reachable through
      :func:`ffpa_attn_amd.hip.forward`, the public API keeps the reference's contract).
  """

  name: str = "hip"
  acc: str = "f32"
  stages: int | None = None
  rescale_threshold: float | None = None

  def __post_init__(self) -> None:
    super().__post_init__()
    if self.acc not in ("f16", "f32"):
      raise AssertionError(f"acc must be 'f16' or 'f32', got {self.acc!r}")
    if self.acc == "f16":
      raise ValueError(
        "HIPBackend(acc='f16') is not available: the gfx950 kernel accumulates in fp32 only "
        "(the reference gates its fp16-acc kernels behind ENABLE_FFPA_F16_ACC as well)."
      )

  @property
  def acc_code(self) -> int:
    return 1


@dataclass
class CUDABackend(HIPBackend):
  """Source-compatible alias: the reference's hand-written native backend
  (functional.py:218-373).  Runs the gfx950 kernel; NVIDIA-only switches are ignored."""

  name: str = "cuda"
  enable_tma: bool | None = None
  enable_cute: bool | None = None
  enable_ws: bool = False
  enable_fp8: bool = False
  enable_fp4: bool = False

  def __post_init__(self) -> None:
    super().__post_init__()
    # the reference's CUDA backend is forward-only: CUDABackend(), backend="cuda" and backward_backend="cuda" trip this very assertion
    # (functional.py:266-268 after Backend.__post_init__ :189-196); callers spell it forward_backend="cuda" / CUDABackend(forward=True)
    assert not self.backward, "cuda backend does not support backward"
    if self.enable_fp8 or self.enable_fp4:
      raise NotImplementedError("FP8 / FP4 attention is out of scope for the MI355X build (bf16/fp16 only)")


@dataclass
class TritonBackend(HIPBackend):
  """Source-compatible alias for the reference's default backend (functional.py:377-421).
  No Triton is used here; the call runs the gfx950 kernel."""

  name: str = "triton"
  enable_ws: bool = False
  autotune: bool = False


@dataclass
class CuTeDSLBackend(Backend):
  """NVIDIA CuTe-DSL backend (functional.py:424-470).  Never available on AMD hardware, so —
  exactly like the reference when ``cute_forward_available()`` is false — it falls back to SDPA."""

  name: str = "cutedsl"


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
