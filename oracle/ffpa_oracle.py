"""Python face of the CPU oracle (``ffpa_oracle.c``) + an fp64 plain-math reference.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module, and only as the checker: the product package
``ffpa_attn_amd`` never does (tests/test_layout_rules.py enforces it).

* :func:`oracle_forward` — the reference's tile recurrence restated in C (see the header of
  ``ffpa_oracle.c`` for the file:line map): log2-domain online softmax, lazy rescale with
  threshold 8, P rounded to the storage dtype before P.V, row sum from unrounded P.
* :func:`math_forward_f64` — ``softmax(scale*QK^T + bias, mask) V`` in float64, no tiling: the
  "what should the answer be" reference used to pin the oracle itself.

Parity pinning: no golden vectors exist in the reference for this path (its tests compare against
PyTorch SDPA on seeded inputs, tests/test_ffpa_fwd.py:106-121).  ``tests/test_oracle.py`` pins this
oracle against committed SDPA-CPU fixtures and against the reference's own ``ffpa_attn_func``
output for config 1, both generated in the authoring container by ``tests/golden/make_golden.py``.
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libffpa_oracle.so")
_lib = None

DEFAULT_THRESHOLD = 8.0  # FFPA_RESCALE_THRESHOLD, csrc/cuffpa/common.cuh:14


def build(force: bool = False) -> str:
  """Compile ``ffpa_oracle.c`` with gcc (a few seconds)."""
  src = os.path.join(_HERE, "ffpa_oracle.c")
  if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
    subprocess.run(
      ["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", src, "-o", _LIB_PATH, "-lm"],
      check=True,
      capture_output=True,
    )
  return _LIB_PATH


def _load() -> ctypes.CDLL:
  global _lib
  if _lib is None:
    build()
    lib = ctypes.CDLL(_LIB_PATH)
    lib.ffpa_oracle_fwd.restype = ctypes.c_int
    lib.ffpa_oracle_fwd.argtypes = [
      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
      ctypes.c_void_p, ctypes.c_void_p,
      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
      ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int,
      ctypes.c_int,
    ]
    lib.ffpa_oracle_fwd_dropout.restype = ctypes.c_int
    lib.ffpa_oracle_fwd_dropout.argtypes = lib.ffpa_oracle_fwd.argtypes + [ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64]
    lib.ffpa_oracle_fwd_ex.restype = ctypes.c_int
    lib.ffpa_oracle_fwd_ex.argtypes = lib.ffpa_oracle_fwd_dropout.argtypes + [ctypes.c_void_p]
    lib.ffpa_oracle_philox.restype = None
    lib.ffpa_oracle_philox.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32)]
    _lib = lib
  return _lib


def philox4x32_10(seed: int, quad: int) -> tuple[int, int, int, int]:
  """Philox4x32-10 block for counter (quad_lo, quad_hi, 0, 0) and key = seed."""
  out = (ctypes.c_uint32 * 4)()
  _load().ffpa_oracle_philox(seed, quad, out)
  return tuple(out)


# ---- 16-bit helpers (numpy has no bfloat16) -------------------------------------------
def bf16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
  return (bits.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
  u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
  rounded = u + 0x7FFF + ((u >> 16) & 1)
  return (rounded >> 16).astype(np.uint16)


def to_bits(x, dtype: str) -> np.ndarray:
  """float array -> uint16 storage bits of ``dtype`` ('bf16' | 'fp16')."""
  if dtype == "bf16":
    return f32_to_bf16_bits(np.asarray(x, dtype=np.float32))
  return np.asarray(x, dtype=np.float16).view(np.uint16)


def from_bits(bits: np.ndarray, dtype: str) -> np.ndarray:
  if dtype == "bf16":
    return bf16_bits_to_f32(bits)
  return bits.view(np.float16).astype(np.float32)


def torch_to_bits(t) -> tuple[np.ndarray, str]:
  """torch bf16/fp16 tensor -> (uint16 numpy bits, dtype name)."""
  import torch

  t = t.detach().cpu().contiguous()
  if t.dtype == torch.bfloat16:
    return t.view(torch.int16).numpy().view(np.uint16), "bf16"
  if t.dtype == torch.float16:
    return t.view(torch.int16).numpy().view(np.uint16), "fp16"
  raise TypeError(f"expected bf16/fp16, got {t.dtype}")


def _bias_strides(bias: np.ndarray, full: tuple[int, int, int, int]) -> np.ndarray:
  assert bias.ndim == 4
  st = []
  for dim in range(4):
    if bias.shape[dim] == full[dim] and full[dim] > 1:
      st.append(bias.strides[dim] // bias.itemsize)
    else:
      assert bias.shape[dim] == 1, f"bias dim {dim}: {bias.shape[dim]} vs {full[dim]}"
      st.append(0)
  return np.asarray(st, dtype=np.int64)


def oracle_forward(
  q_bits: np.ndarray,
  k_bits: np.ndarray,
  v_bits: np.ndarray,
  dtype: str = "bf16",
  *,
  scale: float | None = None,
  causal: bool = False,
  causal_offset: int | None = None,
  bias: np.ndarray | None = None,
  threshold: float = DEFAULT_THRESHOLD,
  block_keys: int = 64,
  rows: tuple[int, int] | None = None,
  dropout_p: float = 0.0,
  philox_seed: int = 0,
  philox_offset: int = 0,
  return_pmax: bool = False,
):
  """Run the C oracle.  Inputs are uint16 storage bits, dense ``[B,H,N,D]``.

  Returns ``(o_bits uint16, o_f32 float32 (unrounded), lse float32)``; with ``rows=(r0, r1)`` only
  those query rows of every (batch, head) are computed (the rest of the outputs is zero).
  ``return_pmax=True`` appends ``pmax float32 [B,Hq,Nq]`` — each row's largest normalised probability (what bounds the
  effect of one differently-rounded P entry on an output element); ``return_pmax="both"`` appends ``(pmax, p2sum)`` with
  ``p2sum = sum_k (P/l)^2`` (what scales the noise the 16-bit rounding of all P entries leaves in an output element).
  """
  lib = _load()
  q_bits = np.ascontiguousarray(q_bits, dtype=np.uint16)
  k_bits = np.ascontiguousarray(k_bits, dtype=np.uint16)
  v_bits = np.ascontiguousarray(v_bits, dtype=np.uint16)
  B, Hq, Nq, D = q_bits.shape
  _, Hkv, Nkv, _ = k_bits.shape
  if scale is None:
    scale = 1.0 / np.sqrt(D)
  if causal_offset is None:
    causal_offset = Nkv - Nq  # tail aligned (split_d.cuh:223)
  o = np.zeros((B, Hq, Nq, D), dtype=np.uint16)
  o32 = np.zeros((B, Hq, Nq, D), dtype=np.float32)
  lse = np.zeros((B, Hq, Nq), dtype=np.float32)
  bias_p, bst_p = None, None
  if bias is not None:
    bias = np.asarray(bias, dtype=np.float32)
    bst = _bias_strides(bias, (B, Hq, Nq, Nkv))
    bias_p, bst_p = bias.ctypes.data, bst.ctypes.data
  r0, r1 = (0, Nq) if rows is None else rows
  pmax = np.zeros((B, Hq, 2, Nq), dtype=np.float32)
  rc = lib.ffpa_oracle_fwd_ex(
    q_bits.ctypes.data, k_bits.ctypes.data, v_bits.ctypes.data, o.ctypes.data, o32.ctypes.data, lse.ctypes.data,
    bias_p, bst_p, B, Hq, Hkv, Nq, Nkv, D, 0 if dtype == "bf16" else 1, float(scale), int(bool(causal)),
    int(causal_offset), float(threshold), int(block_keys), int(r0), int(r1), float(dropout_p), int(philox_seed),
    int(philox_offset), pmax.ctypes.data,
  )
  if rc != 0:
    raise RuntimeError(f"ffpa_oracle_fwd failed ({rc})")
  if return_pmax == "both":
    return o, o32, lse, (pmax[:, :, 0], pmax[:, :, 1])
  if return_pmax:
    return o, o32, lse, pmax[:, :, 0]
  return o, o32, lse


def math_forward_f64(
  q: np.ndarray,
  k: np.ndarray,
  v: np.ndarray,
  *,
  scale: float | None = None,
  causal: bool = False,
  causal_offset: int | None = None,
  bias: np.ndarray | None = None,
):
  """Untiled float64 attention: returns ``(O, LSE)``.  ``q`` ``[B,Hq,Nq,D]``, ``k``/``v``
  ``[B,Hkv,Nkv,D]`` (GQA by head grouping, split_d.cuh:135-136)."""
  q = np.asarray(q, dtype=np.float64)
  k = np.asarray(k, dtype=np.float64)
  v = np.asarray(v, dtype=np.float64)
  B, Hq, Nq, D = q.shape
  _, Hkv, Nkv, _ = k.shape
  g = Hq // Hkv
  if scale is None:
    scale = 1.0 / np.sqrt(D)
  if causal_offset is None:
    causal_offset = Nkv - Nq
  kk = np.repeat(k, g, axis=1)
  vv = np.repeat(v, g, axis=1)
  s = np.einsum("bhqd,bhkd->bhqk", q, kk) * scale
  if bias is not None:
    s = s + np.asarray(bias, dtype=np.float64)
  if causal:
    rows = np.arange(Nq)[:, None]
    keys = np.arange(Nkv)[None, :]
    s = np.where(keys <= rows + causal_offset, s, -np.inf)
  with np.errstate(invalid="ignore", divide="ignore"):
    m = s.max(axis=-1, keepdims=True)
    m_safe = np.where(np.isfinite(m), m, 0.0)
    p = np.exp(s - m_safe)
    l = p.sum(axis=-1, keepdims=True)
    o = np.einsum("bhqk,bhkd->bhqd", p, vv) / l
    lse = (np.log(l) + m_safe)[..., 0]
    lse = np.where(np.isfinite(m[..., 0]), lse, -np.inf)
  return o, lse
