"""``ffpa_attn::_fwd_hip`` — the torch-facing shim over the C-ABI HIP library.

This module is the MI355X replacement for the reference's CUDA op shim
(``src/ffpa_attn/cuda/__init__.py:57-171`` + ``cuda/_ffpa_fwd.py:6-62``):

* it loads ``libffpa_attn_hip.so`` (built in-tree by :mod:`ffpa_attn_amd.build`) with
  ``ctypes`` — the library has no torch dependency, the boundary is
  ``include/ffpa_attn.h``;
* it registers the torch.library op ``ffpa_attn::_fwd_hip`` whose first eleven
  arguments are the reference op's (``q, k, v, attn_bias, stages, acc, causal,
  softmax_scale, dropout_p, philox_seed, philox_offset``) and which returns
  ``(o, softmax_lse)`` with ``softmax_lse`` an exact-length ``[B, Hq, Nq]`` fp32 tensor
  (``cuda/__init__.py:100-112``);
* it registers a fake (meta) implementation so ``torch.compile`` can trace through.

There is NO fallback in here: if the library is missing or was not built for this GPU
the op raises ``RuntimeError`` (the reference raises the same class when ``_C`` was not
compiled, ``cuda/__init__.py:94-99``).
"""

from __future__ import annotations

import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_PKG, "libffpa_attn_hip.so")
TEST_LIB_PATH = os.path.join(_PKG, "libffpa_attn_hip_test.so")  # product kernels + the register-staged SAFE twins (tests only)

ABI_VERSION = 6  # 5: + the packed-sequence entry points (ffpa_attn_varlen_fwd ...); 6: + KV splits inside the packed call

# enum ffpa_status (include/ffpa_attn.h)
_STATUS_EXC = {
  1: RuntimeError,  # NULL pointer
  2: TypeError,  # dtype
  3: RuntimeError,  # "headdim not support!" (env.py:750-752 -> std::runtime_error)
  4: ValueError,  # shape
  5: ValueError,  # stride
  6: ValueError,  # alignment
  7: NotImplementedError,  # unsupported feature
  8: RuntimeError,  # launch failure
  9: RuntimeError,  # no device
  10: RuntimeError,  # ABI mismatch
}

FLAG_DEBUG_SAFE_PATH = 0x1
FLAG_NO_XCD_REMAP = 0x2
FLAG_NO_BIAS_LDS = 0x8
FLAG_L2_PREFETCH = 0x10  # bench-only: touch the K/V tile two steps ahead in every prefill launch (default: the library decides; D > 512)
FLAG_NO_L2_PREFETCH = 0x20  # bench-only: never
FLAG_FORCE_SPLITS = 0x40  # bench-only: honour num_splits > 1 for prefill launches that fill the chip too
FLAG_KV_STREAM = 0x80      # bench-only: force the non-temporal K / V fetch of short-query launches
FLAG_NO_KV_STREAM = 0x800  # bench-only: never
FLAG_WIDE_TILE = 0x1000     # bench / test: prefill launches take the wide-row tile wherever one is built (default: the library decides)
FLAG_NO_WIDE_TILE = 0x2000  # bench / test: never
FLAG_PAIR_TILES = 0x8000      # bench / test: causal prefill launches pair row tiles i and n - 1 - i in one workgroup (default: the library decides)
FLAG_NO_PAIR_TILES = 0x20000  # bench / test: never
FLAG_NO_HEAD_CHUNKS = 0x80000  # bench / test: causal GQA prefill launches keep the (batch, head, row tile) workgroup order (default: the library decides; same bits either way)
FLAG_TILE_RANGES = 0x100000  # bench / test: with FLAG_FORCE_SPLITS and num_splits = n, a causal prefill launch splits every row tile's OWN visible KV tiles into n ranges
FLAG_NO_TILE_RANGES = 0x200000  # bench / test: never (default: causal launches of one round of workgroups take two)
FLAG_NO_COMPACT_GRID = 0x400000  # bench / test, packed-sequence call: keep the grid of batch x ceil(max_seqlen_q / block rows) row tiles per head (default: sized by total_q for ragged prefill batches)
FLAG_NO_PACK_GQA = 0x40000  # bench / test, packed-sequence call: decode batches under GQA keep one workgroup per QUERY head (default: a KV group's heads are the rows of one tile)
FLAG_DETERMINISTIC = 0x4000  # batch-invariant bits: no prefill KV splits, no wide-row tile, short-query splits by the KV length alone (FFPA_HIP_DETERMINISTIC=1 sets it on every call)


def FLAG_XCD_GROUP(n: int) -> int:
  """bench-only: force the number of XCDs (1, 2, 4 or 8) that share a head's row tiles (default: the library decides from the K/V footprint)"""
  return {1: 0x100, 2: 0x200, 4: 0x300, 8: 0x400}[n]

# enum ffpa_bias_dtype: additive fp16 / bf16 / fp32, or a boolean mask read as bytes (non-zero = visible)
# (torch.uint8 is NOT accepted: the public API and the reference take bool / float masks only, functional.py:860-898)
_BIAS_DTYPE = {torch.float16: 1, torch.bfloat16: 2, torch.float32: 3, torch.bool: 4}
_DTYPE = {torch.bfloat16: 0, torch.float16: 1}


class FfpaFwdParams(ctypes.Structure):
  """ctypes mirror of ``struct ffpa_fwd_params`` (include/ffpa_attn.h)."""

  _fields_ = [
    ("struct_size", ctypes.c_uint32),
    ("abi_version", ctypes.c_uint32),
    ("q", ctypes.c_void_p),
    ("k", ctypes.c_void_p),
    ("v", ctypes.c_void_p),
    ("o", ctypes.c_void_p),
    ("lse", ctypes.c_void_p),
    ("bias", ctypes.c_void_p),
    ("batch", ctypes.c_int32),
    ("heads_q", ctypes.c_int32),
    ("heads_kv", ctypes.c_int32),
    ("seqlen_q", ctypes.c_int32),
    ("seqlen_kv", ctypes.c_int32),
    ("head_dim", ctypes.c_int32),
    ("q_stride", ctypes.c_int64 * 3),
    ("k_stride", ctypes.c_int64 * 3),
    ("v_stride", ctypes.c_int64 * 3),
    ("o_stride", ctypes.c_int64 * 3),
    ("bias_stride", ctypes.c_int64 * 4),
    ("dtype", ctypes.c_int32),
    ("bias_dtype", ctypes.c_int32),
    ("causal", ctypes.c_int32),
    ("causal_offset", ctypes.c_int32),
    ("softmax_scale", ctypes.c_float),
    ("rescale_threshold", ctypes.c_float),
    ("dropout_p", ctypes.c_float),
    ("flags", ctypes.c_uint32),
    ("philox_seed", ctypes.c_uint64),
    ("philox_offset", ctypes.c_uint64),
    ("workspace", ctypes.c_void_p),
    ("workspace_bytes", ctypes.c_uint64),
    ("num_splits", ctypes.c_int32),
    ("causal_row_mod", ctypes.c_int32),
    ("kv_bounds", ctypes.c_void_p),
    ("kv_bounds_stride", ctypes.c_int64 * 2),
    ("split_tickets", ctypes.c_void_p),
  ]


class FfpaVarlenFwdParams(ctypes.Structure):
  """ctypes mirror of ``struct ffpa_varlen_fwd_params`` (include/ffpa_attn.h): the packed-sequence call."""

  _fields_ = [
    ("struct_size", ctypes.c_uint32),
    ("abi_version", ctypes.c_uint32),
    ("q", ctypes.c_void_p),
    ("k", ctypes.c_void_p),
    ("v", ctypes.c_void_p),
    ("o", ctypes.c_void_p),
    ("lse", ctypes.c_void_p),
    ("cu_seqlens_q", ctypes.c_void_p),
    ("cu_seqlens_kv", ctypes.c_void_p),
    ("seqused_kv", ctypes.c_void_p),
    ("batch", ctypes.c_int32),
    ("heads_q", ctypes.c_int32),
    ("heads_kv", ctypes.c_int32),
    ("head_dim", ctypes.c_int32),
    ("max_seqlen_q", ctypes.c_int32),
    ("max_seqlen_kv", ctypes.c_int32),
    ("q_stride", ctypes.c_int64 * 2),
    ("k_stride", ctypes.c_int64 * 2),
    ("v_stride", ctypes.c_int64 * 2),
    ("o_stride", ctypes.c_int64 * 2),
    ("lse_stride_head", ctypes.c_int64),
    ("dtype", ctypes.c_int32),
    ("causal", ctypes.c_int32),
    ("softmax_scale", ctypes.c_float),
    ("rescale_threshold", ctypes.c_float),
    ("flags", ctypes.c_uint32),
    ("reserved", ctypes.c_uint32),
    ("workspace", ctypes.c_void_p),
    ("workspace_bytes", ctypes.c_uint64),
    ("num_splits", ctypes.c_int32),
    ("total_q", ctypes.c_int32),
  ]


_lib = None
_debug_lib = None
_lib_lock = threading.Lock()

EXPORTS = (
  "ffpa_attn_fwd",
  "ffpa_attn_fwd_workspace_bytes",
  "ffpa_attn_fwd_split_tickets",
  "ffpa_attn_mask_kv_bounds",
  "ffpa_attn_fwd_plan",
  "ffpa_attn_fwd_kernel",
  "ffpa_attn_varlen_fwd",
  "ffpa_attn_varlen_fwd_plan",
  "ffpa_attn_varlen_fwd_kernel",
  "ffpa_attn_varlen_fwd_workspace_bytes",
  "ffpa_attn_query",
  "ffpa_attn_fwd_tile_config",
  "ffpa_attn_last_error",
  "ffpa_attn_version",
)


def load_library(path: str | None = None) -> ctypes.CDLL:
  """dlopen the C-ABI library (after torch, so both share one HIP runtime) and bind
  every symbol ``include/ffpa_attn.h`` declares.  Raises ``RuntimeError`` if it is missing.
  """
  global _lib
  if _lib is not None and path is None:
    return _lib
  with _lib_lock:
    if _lib is not None and path is None:
      return _lib
    p = path or os.environ.get("FFPA_HIP_LIBRARY") or LIB_PATH  # (FFPA_HIP_LIBRARY: developer override — run the tests against a variant build)
    if not os.path.exists(p):
      raise RuntimeError(
        f"ffpa_attn_amd: {p} not found. The HIP extension is required (there is no fallback "
        "kernel): build it with `python -m ffpa_attn_amd.build` (needs hipcc, targets gfx950)."
      )
    lib = ctypes.CDLL(p)
    lib.ffpa_attn_fwd.argtypes = [ctypes.POINTER(FfpaFwdParams), ctypes.c_void_p]
    lib.ffpa_attn_fwd.restype = ctypes.c_int
    lib.ffpa_attn_mask_kv_bounds.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.ffpa_attn_mask_kv_bounds.restype = ctypes.c_int
    lib.ffpa_attn_fwd_workspace_bytes.argtypes = [ctypes.POINTER(FfpaFwdParams)]
    lib.ffpa_attn_fwd_workspace_bytes.restype = ctypes.c_size_t
    if path is None or hasattr(lib, "ffpa_attn_fwd_split_tickets"):
      lib.ffpa_attn_fwd_split_tickets.argtypes = [ctypes.POINTER(FfpaFwdParams)]
      lib.ffpa_attn_fwd_split_tickets.restype = ctypes.c_size_t
    lib.ffpa_attn_fwd_plan.argtypes = [ctypes.POINTER(FfpaFwdParams), ctypes.POINTER(ctypes.c_int)]
    lib.ffpa_attn_fwd_plan.restype = ctypes.c_int
    if path is None or hasattr(lib, "ffpa_attn_fwd_kernel"):  # (developer A/B runs may load a saved build of an older commit by path)
      lib.ffpa_attn_fwd_kernel.argtypes = [ctypes.POINTER(FfpaFwdParams), ctypes.c_char_p, ctypes.c_size_t]
      lib.ffpa_attn_fwd_kernel.restype = ctypes.c_int
    if path is None or hasattr(lib, "ffpa_attn_varlen_fwd"):  # (ditto: a saved build from before the packed-sequence entry points)
      lib.ffpa_attn_varlen_fwd.argtypes = [ctypes.POINTER(FfpaVarlenFwdParams), ctypes.c_void_p]
      lib.ffpa_attn_varlen_fwd.restype = ctypes.c_int
      lib.ffpa_attn_varlen_fwd_plan.argtypes = [ctypes.POINTER(FfpaVarlenFwdParams), ctypes.POINTER(ctypes.c_int)]
      lib.ffpa_attn_varlen_fwd_plan.restype = ctypes.c_int
      lib.ffpa_attn_varlen_fwd_workspace_bytes.argtypes = [ctypes.POINTER(FfpaVarlenFwdParams)]
      lib.ffpa_attn_varlen_fwd_workspace_bytes.restype = ctypes.c_size_t
      lib.ffpa_attn_varlen_fwd_kernel.argtypes = [ctypes.POINTER(FfpaVarlenFwdParams), ctypes.c_char_p, ctypes.c_size_t]
      lib.ffpa_attn_varlen_fwd_kernel.restype = ctypes.c_int
    lib.ffpa_attn_query.argtypes = [ctypes.c_int]
    lib.ffpa_attn_query.restype = ctypes.c_int
    lib.ffpa_attn_fwd_tile_config.argtypes = [
      ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)
    ]
    lib.ffpa_attn_fwd_tile_config.restype = ctypes.c_int
    lib.ffpa_attn_last_error.argtypes = []
    lib.ffpa_attn_last_error.restype = ctypes.c_char_p
    lib.ffpa_attn_version.argtypes = []
    lib.ffpa_attn_version.restype = ctypes.c_char_p
    if lib.ffpa_attn_query(0) != ABI_VERSION and path is None:
      raise RuntimeError(f"ffpa_attn_amd: {p} has ABI {lib.ffpa_attn_query(0)}, expected {ABI_VERSION}")
    if path is None:
      _lib = lib
    return lib


def load_debug_library() -> ctypes.CDLL:
  """The test-only twin of the library (``libffpa_attn_hip_test.so``: the same kernels plus the register-staged
  SAFE variants behind ``FLAG_DEBUG_SAFE_PATH``).  The product library does not carry them
  (``ffpa_attn_query(FFPA_QUERY_DEBUG_KERNELS) == 0``)."""
  global _debug_lib
  if _debug_lib is None:
    _debug_lib = load_library(TEST_LIB_PATH)
  return _debug_lib


def library_available() -> bool:
  return os.path.exists(LIB_PATH)


# Module-level capability attributes, under the names the reference's CUDA shim exports at import (``src/ffpa_attn/cuda/__init__.py:6-25``:
# read from its pybind module, ``csrc/cuffpa/ffpa_api.cc:283-305``) — call sites that gate on ``ffpa_attn.cuda.CUDA_FWD_AVAILABLE`` keep working
# against this module.  Answered lazily from ``ffpa_attn_query()`` (PEP 562: importing the package must not need the built library).  The
# reference's process-global ``set/get_cuda_backend_impl`` hint (``:38-47``, ``backend.h:16-27``) has NO equivalent on purpose: the C-ABI keeps
# no mutable global state, every choice travels in ``ffpa_fwd_params`` (SURVEY.md section 8b "Threading").
_CAPABILITY_QUERIES = {
  "HIP_FWD_AVAILABLE": 1, "CUDA_FWD_AVAILABLE": 1,  # FFPA_QUERY_FWD_AVAILABLE
  "VARLEN_FWD_AVAILABLE": 11,                        # FFPA_QUERY_VARLEN_AVAILABLE (ffpa_attn_varlen_func: the reference needs its CuTe-DSL backend for it)
  "FP16_AVAILABLE": 5, "DROPOUT_AVAILABLE": 6,       # FFPA_QUERY_FP16_AVAILABLE / _DROPOUT_AVAILABLE
}
_CAPABILITY_CONSTANTS = {
  "F16_ACC_AVAILABLE": False,        # fp32 accumulation only (the reference builds its fp16-acc kernels behind ENABLE_FFPA_F16_ACC)
  "CUDA_TMA_AVAILABLE": False,       # sm_90+ hardware feature
  "CUDA_CUTE_TMA_AVAILABLE": False,  # sm_120 CuTe kernels
  "CUDA_BWD_AVAILABLE": False,       # as in the reference: the native backend is forward-only
}


def __getattr__(name: str):
  if name == "DecodeStep":  # (the graph-replayed decode step lives above the op: ffpa_attn_amd/decode.py; reachable as hip.DecodeStep too)
    from ..decode import DecodeStep

    return DecodeStep
  if name in _CAPABILITY_CONSTANTS:
    return _CAPABILITY_CONSTANTS[name]
  if name in _CAPABILITY_QUERIES:
    try:
      return load_library().ffpa_attn_query(_CAPABILITY_QUERIES[name]) == 1
    except (RuntimeError, OSError):
      return False  # not built: the reference answers False when its extension module is missing, too
  raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def tile_config(head_dim: int) -> dict:
  """Rows per workgroup / keys per tile / LDS bytes the kernel uses for ``head_dim``."""
  lib = load_library()
  br, bc, lds = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
  rc = lib.ffpa_attn_fwd_tile_config(int(head_dim), ctypes.byref(br), ctypes.byref(bc), ctypes.byref(lds))
  if rc != 0:
    raise _STATUS_EXC.get(rc, RuntimeError)(lib.ffpa_attn_last_error().decode())
  return {"block_rows": br.value, "block_keys": bc.value, "lds_bytes": lds.value}


def launch_plan(batch: int, heads_q: int, heads_kv: int, seqlen_q: int, seqlen_kv: int, head_dim: int, *, dtype: torch.dtype = torch.bfloat16,
                causal: bool = False, bias_dtype: "torch.dtype | None" = None, dropout_p: float = 0.0, flags: int = 0, num_splits: int = 0,
                device: "torch.device | int | None" = None) -> dict:
  """The launch plan the library would pick for a call of this shape class — tile (``block_rows`` x ``block_keys``), KV ``splits``, ``variant``
  (0 prefill, 1 short-query) and the ``kernel`` name — without launching anything (``ffpa_attn_fwd_plan`` / ``_kernel`` on a parameter block with
  placeholder pointers; with ``device`` the plan is that GPU's: its CU count prices the split rules).  A caller that cuts a batch into pieces and needs the
  pieces to run the plan of the whole (``sharding.attend_and_gather_units``) asks here."""
  lib = load_library()
  d8 = (int(head_dim) + 7) // 8 * 8
  p = FfpaFwdParams()
  p.struct_size = ctypes.sizeof(FfpaFwdParams)
  p.abi_version = ABI_VERSION
  p.q = p.k = p.v = p.o = 16
  p.batch, p.heads_q, p.heads_kv, p.seqlen_q, p.seqlen_kv, p.head_dim = int(batch), int(heads_q), int(heads_kv), int(seqlen_q), int(seqlen_kv), d8
  for name, h, n in (("q_stride", heads_q, seqlen_q), ("k_stride", heads_kv, seqlen_kv), ("v_stride", heads_kv, seqlen_kv), ("o_stride", heads_q, seqlen_q)):
    getattr(p, name)[:] = [h * n * d8, n * d8, d8]
  p.dtype = _DTYPE[dtype]
  p.causal = 1 if causal else 0
  p.causal_offset = int(seqlen_kv) - int(seqlen_q)
  p.softmax_scale = float(head_dim) ** -0.5
  p.rescale_threshold = -1.0
  p.dropout_p = float(dropout_p)
  p.flags = int(flags)
  p.num_splits = int(num_splits)
  if bias_dtype is not None:
    p.bias = 16
    p.bias_dtype = _BIAS_DTYPE[bias_dtype]
    p.bias_stride[:] = [0, 0, int(seqlen_kv), 1]
  if num_splits != 1:
    p.workspace, p.workspace_bytes = 16, (1 << 62)
  plan = (ctypes.c_int * 4)()
  name = ctypes.create_string_buffer(160)

  def ask():
    rc = lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan)
    if rc != 0:
      raise _STATUS_EXC.get(rc, RuntimeError)(lib.ffpa_attn_last_error().decode())
    lib.ffpa_attn_fwd_kernel(ctypes.byref(p), name, len(name))

  if device is not None and torch.cuda.is_available():
    with torch.cuda.device(device):
      ask()
  else:
    ask()
  return {"variant": plan[0], "block_rows": plan[1], "block_keys": plan[2], "splits": plan[3], "kernel": name.value.decode()}


def padded_head_dim(d: int) -> int:
  """The head dim of the kernel instantiation that serves ``d``: kernels are built per multiple of 64; a head dim in
  between (any multiple of 8) runs on the next one with the missing columns read as zeros and never stored — in the
  kernel, without padded copies (the reference pads on the host, csrc/cuffpa/ffpa_api.cc:123-161)."""
  return ((d + 63) // 64) * 64


def _dense_rows(t: torch.Tensor) -> torch.Tensor:
  """Make ``t`` satisfy the layout contract: headdim stride 1, all other strides multiples of
  8 elements, 16-byte aligned base (split_d.cuh:137-142 assumed dense [B,H,N,D]; here arbitrary
  batch/head/row strides are honoured and only pathological views are copied)."""
  ok = t.stride(-1) == 1 and all(s % 8 == 0 for s in t.stride()[:-1]) and t.data_ptr() % 16 == 0
  if ok and t.dim() == 4 and t.size(2) > 1 and t.stride(2) < t.size(3):
    ok = False  # overlapping rows
  return t if ok else t.contiguous()


def mask_kv_bounds(attn_bias: torch.Tensor, nq: int, nkv: int) -> torch.Tensor:
  """Key ranges of an additive (-inf = hidden, 0 = neutral) or boolean (False = hidden) mask for the kernel's tile clipping:
  int32 ``[Bb, Hb, ceil(Nq/32), 4]`` holding, per block of 32 query rows, ``[first, end)`` such that every key outside is
  hidden for every row of the block (``{Nkv, 0}`` for a block without any visible key) and ``[free_first, free_end)``, the
  first run of keys on which the mask does nothing for every row of the block (``{0, 0}``: none) — tiles inside it are
  computed without reading the mask.  On the GPU: one fused pass of the library over the mask
  (``ffpa_attn_mask_kv_bounds``); elsewhere (tests) the same thing in torch ops."""
  bb, hb = attn_bias.size(0), attn_bias.size(1)
  nblk = (nq + 31) // 32
  if attn_bias.is_cuda and attn_bias.dtype in _BIAS_DTYPE:
    lib = load_library()
    out = torch.empty((bb, hb, nblk, 4), dtype=torch.int32, device=attn_bias.device)
    strides = (ctypes.c_int64 * 4)(*[attn_bias.stride(d) if attn_bias.size(d) > 1 else 0 for d in range(4)])
    with torch.cuda.device(attn_bias.device):
      rc = lib.ffpa_attn_mask_kv_bounds(ctypes.c_void_p(attn_bias.data_ptr()), _BIAS_DTYPE[attn_bias.dtype], strides, bb, hb, nq, nkv,
                                        ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream(attn_bias.device).cuda_stream))
    if rc != 0:
      raise _STATUS_EXC.get(rc, RuntimeError)(lib.ffpa_attn_last_error().decode())
    return out
  is_bool = attn_bias.dtype in (torch.bool, torch.uint8)
  vis = attn_bias.ne(0) if is_bool else ~torch.isneginf(attn_bias)      # [Bb, Hb, Nq|1, Nkv|1]
  neutral = attn_bias.ne(0) if is_bool else attn_bias.eq(0)
  pad = nblk * 32 - nq

  def per_block(t, pad_value, reduce):
    t = t.expand(bb, hb, nq, nkv)
    if pad:
      t = torch.nn.functional.pad(t, (0, 0, 0, pad), value=pad_value)  # rows past Nq: see nothing / constrain nothing
    t = t.reshape(bb, hb, nblk, 32, nkv)
    return t.any(dim=3) if reduce == "any" else t.all(dim=3)            # [Bb, Hb, nblk, Nkv]

  col_any, col_all = per_block(vis, False, "any"), per_block(neutral, True, "all")
  idx = torch.arange(nkv, device=attn_bias.device, dtype=torch.int32)
  first = torch.where(col_any, idx, torch.full_like(idx, nkv)).amin(dim=-1)
  end = torch.where(col_any, idx + 1, torch.zeros_like(idx)).amax(dim=-1)
  ff = torch.where(col_all, idx, torch.full_like(idx, nkv)).amin(dim=-1)                           # first neutral key
  after = (~col_all) & (idx >= ff.unsqueeze(-1))
  fe = torch.where(after, idx, torch.full_like(idx, nkv)).amin(dim=-1)                              # first non-neutral key after it
  none = ff >= nkv
  ff, fe = torch.where(none, torch.zeros_like(ff), ff), torch.where(none, torch.zeros_like(fe), fe)
  return torch.stack((first, end, ff, fe), dim=-1).to(torch.int32).contiguous()


# One scan per mask instead of one per call — OPT-IN (FFPA_HIP_MASK_BOUNDS_CACHE=1): a static mask (the same causal / padding /
# sliding-window tensor for every layer and step) is then scanned once.  An entry is valid only while the very tensor object it was
# computed from (the view's base: the public API re-views the caller's mask on every call) is still alive — so its memory cannot have
# been recycled for another mask — and torch's in-place version counter (shared by all views of a storage, bumped by every in-place
# write torch knows of) has not moved.  What the counter does NOT see leaves stale ranges behind, and stale ranges make the kernel skip
# visible keys silently: `mask.data.fill_()`, a Triton / custom kernel writing through the raw pointer, a graph replay refreshing a
# static mask buffer.  Hence off unless the caller asks for it and knows its masks are written through torch ops only.  Tensors
# without a version counter (created under torch.inference_mode()) are never cached.  An entry remembers the stream its scan ran
# on and an event behind it: a consumer on another stream waits for that event first.  Holds the small int32 result only.
_BOUNDS_CACHE: "dict[tuple, tuple]" = {}
_BOUNDS_CACHE_MAX = 16


def _mask_bounds_cache_enabled() -> bool:
  return os.environ.get("FFPA_HIP_MASK_BOUNDS_CACHE", "0") not in ("0", "")


def _tensor_version(t: torch.Tensor):
  """torch's in-place version counter of ``t``, or None for tensors that do not track one (inference tensors raise on access)."""
  try:
    if t.is_inference():
      return None
    return t._version
  except RuntimeError:
    return None


def cached_mask_kv_bounds(attn_bias: torch.Tensor, nq: int, nkv: int) -> torch.Tensor:
  import weakref

  version = _tensor_version(attn_bias)
  if version is None:  # no version counter: nothing would ever invalidate the entry
    return mask_kv_bounds(attn_bias, nq, nkv)
  owner = attn_bias._base if attn_bias._base is not None else attn_bias
  key = (id(owner), attn_bias.data_ptr(), attn_bias.dtype, tuple(attn_bias.shape), tuple(attn_bias.stride()), nq, nkv)
  stream = torch.cuda.current_stream(attn_bias.device)
  hit = _BOUNDS_CACHE.get(key)
  if hit is not None:
    ref, ver, out, src_stream, event = hit
    if ref() is owner and ver == version:
      if src_stream != stream.cuda_stream:
        stream.wait_event(event)  # the scan ran on another stream: order this stream's launch behind it
      return out
  out = mask_kv_bounds(attn_bias, nq, nkv)
  event = torch.cuda.Event()
  event.record(stream)
  for k_ in [k_ for k_, ent in _BOUNDS_CACHE.items() if ent[0]() is None]:  # owners that died
    del _BOUNDS_CACHE[k_]
  if len(_BOUNDS_CACHE) >= _BOUNDS_CACHE_MAX:
    _BOUNDS_CACHE.pop(next(iter(_BOUNDS_CACHE)))
  _BOUNDS_CACHE[key] = (weakref.ref(owner), version, out, stream.cuda_stream, event)
  return out


def _want_mask_bounds(attn_bias: torch.Tensor, b: int, hq: int, nq: int, nkv: int) -> bool:
  """The scan reads the mask once (2-4 B per element): worth it for every mask that has real row and key axes and a
  problem with tiles to skip, unless each (batch, head) pair brings its own mask (then the scan is as big as the
  mask reads of the attention itself).  FFPA_HIP_MASK_BOUNDS=0/1 forces it off / on."""
  env = os.environ.get("FFPA_HIP_MASK_BOUNDS")
  if env is not None:
    return env not in ("0", "")
  if attn_bias.size(2) != nq or attn_bias.size(3) != nkv or nkv < 512 or nq < 128:
    return False
  return b * hq >= 2 * attn_bias.size(0) * attn_bias.size(1)


# Per-(device, stream) scratch of the KV-split launches.  Launches on one stream are ordered, so they may share a buffer; launches on different
# streams may overlap and get their own.  Both tables are small LRU maps (a process that churns through streams does not accumulate buffers).
#   _TICKETS:    zeroed int32 counters for the in-launch merge (ffpa_fwd_params.split_tickets): zeroed once — the kernel leaves every counter at
#                zero — and grown on demand;
#   _WORKSPACES: the fp32 partials + LSE of the short-query (decode) launches — a fresh torch.empty per token was a fifth of the host time of a
#                decode step.  Only workspaces up to _WORKSPACE_KEEP_BYTES are kept; the transient hundreds of MiB of a split PREFILL launch go
#                back to the caching allocator as before.
_SCRATCH_MAX_STREAMS = 32
_WORKSPACE_KEEP_BYTES = 8 << 20
_TICKETS: "dict[tuple, torch.Tensor]" = {}
_WORKSPACES: "dict[tuple, torch.Tensor]" = {}


def _lru_get(table: dict, key):
  t = table.pop(key, None)
  if t is not None:
    table[key] = t  # (re-inserted: dicts keep insertion order, the first key is the least recently used)
  return t


def _lru_put(table: dict, key, value) -> None:
  table.pop(key, None)
  while len(table) >= _SCRATCH_MAX_STREAMS:
    table.pop(next(iter(table)))
  table[key] = value


def _split_tickets(device: torch.device, stream: int, n: int) -> torch.Tensor:
  if torch.cuda.is_current_stream_capturing():
    # inside a stream capture: a buffer of the graph's own pool, zeroed by a node of the graph (like the workspace, it is replayed in place)
    return torch.zeros(n, dtype=torch.int32, device=device)
  key = (device.index, stream)
  t = _lru_get(_TICKETS, key)
  if t is None or t.numel() < n:
    t = torch.zeros(max(4096, n), dtype=torch.int32, device=device)
    _lru_put(_TICKETS, key, t)
  return t


def _workspace(device: torch.device, stream: int, nbytes: int) -> torch.Tensor:
  words = (nbytes + 3) // 4
  if nbytes > _WORKSPACE_KEEP_BYTES or torch.cuda.is_current_stream_capturing():
    return torch.empty(words, dtype=torch.float32, device=device)  # caching allocator (under capture: the graph's pool, replayed in place)
  key = (device.index, stream)
  t = _lru_get(_WORKSPACES, key)
  if t is None or t.numel() < words:
    t = torch.empty(max(words, 1 << 16), dtype=torch.float32, device=device)
    _lru_put(_WORKSPACES, key, t)
  return t


# What a launch needs besides its tensors — workspace bytes and ticket count — is a function of the shape class only (ffpa_capi.hip make_plan): asked
# once per class, not once per call (two ctypes round trips less per decode token).  The class names everything make_plan prices with (the mask KIND too: the
# wide-row tile of D = 320 is for boolean masks only, and another tile is another number of workgroups); the library clamps the split count to the scratch it is
# handed, so a class missing from the key could cost splits, never memory safety.
_PLAN_SCRATCH: "dict[tuple, tuple[int, int]]" = {}


def _plan_scratch(lib, p: "FfpaFwdParams", num_splits: int, want_tickets: bool, device_index: int) -> tuple[int, int]:
  if num_splits == 1:
    return 0, 0
  key = (id(lib), device_index, p.dtype, p.batch, p.heads_q, p.heads_kv, p.seqlen_q, p.seqlen_kv, p.head_dim, p.causal, p.bias_dtype if p.bias else 0,
         bool(p.bias) and p.bias_stride[2] == 0, p.kv_bounds is not None, p.dropout_p > 0.0, p.flags, num_splits, p.causal_row_mod, p.causal_offset, want_tickets, os.environ.get("FFPA_HIP_FAKE_CUS"))
  hit = _PLAN_SCRATCH.get(key)
  if hit is None:
    ws = int(lib.ffpa_attn_fwd_workspace_bytes(ctypes.byref(p)))
    nt = int(lib.ffpa_attn_fwd_split_tickets(ctypes.byref(p))) if (ws and want_tickets and hasattr(lib, "ffpa_attn_fwd_split_tickets")) else 0
    if len(_PLAN_SCRATCH) >= 512:
      _PLAN_SCRATCH.clear()
    hit = _PLAN_SCRATCH[key] = (ws, nt)
  return hit


def forward(
  q: torch.Tensor,
  k: torch.Tensor,
  v: torch.Tensor,
  attn_bias: torch.Tensor | None,
  causal: bool,
  softmax_scale: float,
  *,
  causal_offset: int | None = None,
  rescale_threshold: float = -1.0,
  dropout_p: float = 0.0,
  philox_seed: int = 0,
  philox_offset: int = 0,
  flags: int = 0,
  return_lse: bool = True,
  num_splits: int = 0,
  plan_out: dict | None = None,
  kv_bounds: torch.Tensor | bool | None = None,
  causal_row_mod: int = 0,
  merge_in_launch: bool | None = None,
) -> tuple[torch.Tensor, torch.Tensor | None]:
  """Run the gfx950 kernel on the current stream of ``q.device``; returns ``(o, lse)``.

  Inputs are ``[B, H, N, D]`` bf16/fp16 device tensors.  ``causal_offset=None`` selects the
  reference's tail-aligned causal mask (``Nkv - Nq``, split_d.cuh:222-228).

  Short-query calls (the reference's decode regime, ``Nq`` in 1..7 reaches FFPA: functional.py:722-723):
  with GQA and no bias, the ``group`` query heads of a KV head are packed into the row axis
  (``q`` viewed as ``[B, Hkv, group*Nq, D]`` — K/V are then streamed once per KV head instead of once
  per query head), the library splits the KV axis over workgroups (``num_splits``: 0 = heuristic,
  1 = never) and merges by LSE; the scratch for the partials is allocated here with torch.
  ``plan_out``, if given, receives the launch plan (variant, block_rows, block_keys, splits, packed).

  ``merge_in_launch``: short-query KV-split launches merge their partials inside the launch (the last split of a row tile to arrive does
  it: one launch per call) instead of launching the merge kernel behind the split kernel — the same numbers.  OFF by default
  (``FFPA_HIP_MERGE_IN_LAUNCH=1`` or ``True`` turns it on): measured on MI355X the hand-off costs every split workgroup more than
  the second launch costs the call (decode B1 H32 Nkv 8192 D512: 115 vs 110 us; profiles/r03_split_merge.txt).

  ``kv_bounds``: key ranges of the mask (``mask_kv_bounds``) — the kernel then skips the KV tiles the mask hides
  entirely and does not read the mask for the tiles it leaves untouched (an explicit causal mask costs what ``is_causal``
  costs); ``None`` derives them from ``attn_bias``
  when that is worth a scan of the mask, ``False`` never, ``True`` always, or pass a precomputed tensor.  The scan runs on
  every call; ``FFPA_HIP_MASK_BOUNDS_CACHE=1`` keeps the ranges per (mask tensor, torch version counter) instead — only safe when
  the mask is written through torch ops alone (``mask.data`` writes, raw-pointer kernels and graph replays do not move the counter and
  would leave stale ranges, i.e. silently skipped keys); masks created under ``torch.inference_mode()`` are never cached.
  """
  if not q.is_cuda:
    raise NotImplementedError(
      f"ffpa_attn::_fwd_hip has no implementation for device '{q.device.type}' (the HIP kernel needs a GPU tensor)"
    )
  lib = load_debug_library() if (flags & FLAG_DEBUG_SAFE_PATH) else load_library()
  if q.dtype not in _DTYPE or k.dtype != q.dtype or v.dtype != q.dtype:
    raise TypeError(f"ffpa_attn::_fwd_hip only supports fp16/bf16 q/k/v of one dtype, got {q.dtype}, {k.dtype}, {v.dtype}")
  # The kernel takes raw pointers: everything its buffer descriptors assume is checked here, as the reference's
  # launcher does with TORCH_CHECK (csrc/cuffpa/launch.cuh:79-129) — the public API validates earlier, but the
  # registered op can be called directly.
  if q.dim() != 4 or k.dim() != 4 or v.dim() != 4:
    raise ValueError("ffpa_attn::_fwd_hip: q/k/v must be 4-D [B, H, N, D] tensors")
  if k.shape != v.shape:
    raise ValueError(f"ffpa_attn::_fwd_hip: key and value must have the same shape, got {tuple(k.shape)} and {tuple(v.shape)}")
  if k.size(0) != q.size(0) or k.size(3) != q.size(3):
    raise ValueError(f"ffpa_attn::_fwd_hip: q {tuple(q.shape)} and k/v {tuple(k.shape)} must share batch size and head dim")
  if k.size(1) == 0 or q.size(1) % k.size(1) != 0:
    raise ValueError(f"ffpa_attn::_fwd_hip: query num_heads ({q.size(1)}) must be a multiple of key/value num_heads ({k.size(1)})")
  if k.device != q.device or v.device != q.device:
    raise ValueError(f"ffpa_attn::_fwd_hip: q/k/v must be on one device, got {q.device}, {k.device}, {v.device}")
  if attn_bias is not None and attn_bias.numel() > 0 and attn_bias.device != q.device:
    raise ValueError(f"ffpa_attn::_fwd_hip: attn_bias must be on q's device, got {attn_bias.device} and {q.device}")
  B, Hq, Nq, D = q.shape
  _, Hkv, Nkv, _ = k.shape
  out_shape = (B, Hq, Nq)
  if causal_offset is None:
    causal_offset = Nkv - Nq
  Dp = (D + 7) // 8 * 8  # rows must be whole 16-byte slots: only a head dim that is not a multiple of 8 is padded (copies)
  if Dp != D:
    pad = (0, Dp - D)
    q, k, v = (torch.nn.functional.pad(t, pad) for t in (q, k, v))
  group = Hq // Hkv if Hkv and Hq % Hkv == 0 else 1
  has_bias = attn_bias is not None and attn_bias.numel() > 0
  packed = causal_row_mod == 0 and group > 1 and Nq <= 7 and group * Nq <= 32 and not has_bias and dropout_p == 0.0
  if packed:
    # [B, Hq, Nq, D] -> [B, Hkv, group*Nq, D]: packed row r = (head in group) * Nq + (query row)
    q = q.contiguous().view(B, Hkv, group * Nq, Dp)
    causal_row_mod = Nq
    Hq, Nq = Hkv, group * Nq
  q, k, v = _dense_rows(q), _dense_rows(k), _dense_rows(v)
  o = torch.empty((B, Hq, Nq, Dp), dtype=q.dtype, device=q.device)
  lse = torch.empty((B, Hq, Nq), dtype=torch.float32, device=q.device) if return_lse else None

  p = FfpaFwdParams()
  p.struct_size = ctypes.sizeof(FfpaFwdParams)
  p.abi_version = ABI_VERSION
  p.q, p.k, p.v, p.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
  p.lse = lse.data_ptr() if lse is not None else None
  p.batch, p.heads_q, p.heads_kv = B, Hq, Hkv
  p.seqlen_q, p.seqlen_kv, p.head_dim = Nq, Nkv, Dp
  for name, t in (("q_stride", q), ("k_stride", k), ("v_stride", v), ("o_stride", o)):
    getattr(p, name)[:] = list(t.stride()[:3])
  if attn_bias is not None and attn_bias.numel() > 0:
    if attn_bias.dim() != 4:
      raise ValueError("attn_bias must be 4-D [B|1, Hq|1, Nq|1, Nkv|1]")
    if attn_bias.dtype not in _BIAS_DTYPE:
      raise TypeError(f"attn_bias dtype must be fp16/bf16/fp32 (additive) or bool (mask), got {attn_bias.dtype}")
    full = (B, Hq, Nq, Nkv)
    strides = []
    for dim in range(4):
      if attn_bias.size(dim) == full[dim]:
        strides.append(attn_bias.stride(dim) if full[dim] > 1 else 0)
      elif attn_bias.size(dim) == 1:
        strides.append(0)  # broadcast dims use stride 0 (native/launch.cuh:277-290)
      else:
        raise ValueError(f"attn_bias dim {dim} must be 1 or {full[dim]}, got {attn_bias.size(dim)}")
    p.bias = attn_bias.data_ptr()
    p.bias_dtype = _BIAS_DTYPE[attn_bias.dtype]
    p.bias_stride[:] = strides
    if kv_bounds is True or (kv_bounds is None and _want_mask_bounds(attn_bias, B, Hq, Nq, Nkv)):
      # (the scan is skipped under stream capture / compile tracing only in the sense that the cache is not consulted there: a
      # captured graph must contain the scan kernel it depends on)
      capturing = torch.cuda.is_current_stream_capturing()
      kv_bounds = cached_mask_kv_bounds(attn_bias, Nq, Nkv) if (_mask_bounds_cache_enabled() and not capturing) else mask_kv_bounds(attn_bias, Nq, Nkv)
    if isinstance(kv_bounds, torch.Tensor):
      nblk = (Nq + 31) // 32
      if kv_bounds.dtype != torch.int32 or kv_bounds.dim() != 4 or kv_bounds.shape[2:] != (nblk, 4) or not kv_bounds.is_contiguous():
        raise ValueError(f"kv_bounds must be a contiguous int32 [B|1, Hq|1, {nblk}, 4] tensor (mask_kv_bounds)")
      if kv_bounds.size(0) not in (1, B) or kv_bounds.size(1) not in (1, Hq) or kv_bounds.device != q.device:
        raise ValueError("kv_bounds batch / head dims must be 1 or match q, on q's device")
      p.kv_bounds = kv_bounds.data_ptr()
      p.kv_bounds_stride[:] = [kv_bounds.stride(0) if kv_bounds.size(0) > 1 else 0, kv_bounds.stride(1) if kv_bounds.size(1) > 1 else 0]
  else:
    p.bias = None
    p.bias_dtype = 0
  if num_splits == 0 and Nq > 32 and os.environ.get("FFPA_HIP_PREFILL_SPLITS", "1").lower() in ("0", "off", "false", "no"):
    # opt-out of the KV-split rules for PREFILL launches (under-filled / part of a round / ragged round: ffpa_capi.hip make_plan): they allocate fp32
    # scratch of splits x B x Hq x Nq x (D + 1) x 4 bytes per call (up to kMaxAutoWorkspaceBytes = 1 GiB) and make the bits of a (batch, head) slice depend on
    # how many heads share the launch (fp32 partials + LSE merge: equal to rounding, not to the bit).  Short-query (decode) launches keep their rule.
    num_splits = 1
  p.dtype = _DTYPE[q.dtype]
  p.causal = 1 if causal else 0
  p.causal_offset = int(causal_offset)
  p.causal_row_mod = int(causal_row_mod)
  p.num_splits = int(num_splits)
  p.softmax_scale = float(softmax_scale)
  p.rescale_threshold = float(rescale_threshold)
  p.dropout_p = float(dropout_p)
  p.flags = int(flags) | (FLAG_DETERMINISTIC if os.environ.get("FFPA_HIP_DETERMINISTIC", "0").lower() not in ("0", "", "off", "false", "no") else 0)
  p.philox_seed = int(philox_seed) & 0xFFFFFFFFFFFFFFFF
  p.philox_offset = int(philox_offset) & 0xFFFFFFFFFFFFFFFF

  tickets = None
  with torch.cuda.device(q.device):
    stream = torch.cuda.current_stream(q.device).cuda_stream
    if merge_in_launch is None:
      merge_in_launch = os.environ.get("FFPA_HIP_MERGE_IN_LAUNCH", "0") not in ("0", "")
    ws_bytes, n_tickets = _plan_scratch(lib, p, num_splits, bool(merge_in_launch), q.device.index or 0)
    workspace = None
    if ws_bytes:
      workspace = _workspace(q.device, stream, ws_bytes)  # (held in a local until the launch below has been enqueued)
      p.workspace = workspace.data_ptr()
      p.workspace_bytes = ws_bytes
      if n_tickets:
        tickets = _split_tickets(q.device, stream, n_tickets)
        p.split_tickets = tickets.data_ptr()
    if plan_out is not None:
      plan = (ctypes.c_int * 4)()
      if lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan) == 0:
        plan_out.update(variant=plan[0], block_rows=plan[1], block_keys=plan[2], splits=plan[3], packed=bool(packed))
      name = ctypes.create_string_buffer(160)
      if hasattr(lib, "ffpa_attn_fwd_kernel") and lib.ffpa_attn_fwd_kernel(ctypes.byref(p), name, len(name)) == 0:
        plan_out["kernel"] = name.value.decode()
    rc = lib.ffpa_attn_fwd(ctypes.byref(p), ctypes.c_void_p(stream))
  if rc != 0:
    if tickets is not None:
      tickets.zero_()  # a launch that did not complete may have left counters behind: the next call must start from zeros
    raise _STATUS_EXC.get(rc, RuntimeError)(f"ffpa_attn_fwd: {lib.ffpa_attn_last_error().decode()} (status {rc})")
  del tickets
  if packed:
    o = o.view(*out_shape, Dp)
    lse = lse.view(out_shape) if lse is not None else None
  if Dp != D:
    o = o[..., :D].contiguous()  # the op's contract (and its fake impl): a dense [B, Hq, Nq, D]
  return o, lse


# ----------------------------------------------------------------------------------
# torch.library op.  Same leading schema as ffpa_attn::_fwd_cuda
# (src/ffpa_attn/cuda/__init__.py:57-66); the fp8/fp4 tail of that schema is dropped
# (bf16/fp16 only) and two trailing knobs are added with defaults.
# ----------------------------------------------------------------------------------
_OP_NAMESPACE = "ffpa_attn"

torch.library.define(
  f"{_OP_NAMESPACE}::_fwd_hip",
  "(Tensor q, Tensor k, Tensor v, Tensor attn_bias, int stages, int acc, int causal, "
  "float softmax_scale, float dropout_p, int philox_seed, int philox_offset, "
  "int causal_offset=-2147483648, float rescale_threshold=-1.0, Tensor? kv_bounds=None) -> (Tensor o, Tensor softmax_lse)",
)

_AUTO_OFFSET = -2147483648


@torch.library.impl(f"{_OP_NAMESPACE}::_fwd_hip", "CUDA")  # ROCm tensors dispatch on the CUDA key
def _fwd_hip_torch_op(
  q,
  k,
  v,
  attn_bias,
  stages,
  acc,
  causal,
  softmax_scale,
  dropout_p,
  philox_seed,
  philox_offset,
  causal_offset=_AUTO_OFFSET,
  rescale_threshold=-1.0,
  kv_bounds=None,
):
  del stages, acc  # tile/pipeline shape is fixed per head dim; accumulation is always fp32
  o, lse = forward(
    q,
    k,
    v,
    attn_bias if attn_bias.numel() > 0 else None,
    bool(causal),
    softmax_scale,
    causal_offset=None if causal_offset == _AUTO_OFFSET else causal_offset,
    rescale_threshold=rescale_threshold,
    dropout_p=dropout_p,
    philox_seed=philox_seed,
    philox_offset=philox_offset,
    kv_bounds=kv_bounds,
  )
  return o, lse


@torch.library.register_fake(f"{_OP_NAMESPACE}::_fwd_hip")
def _fwd_hip_fake(
  q,
  k,
  v,
  attn_bias,
  stages,
  acc,
  causal,
  softmax_scale,
  dropout_p,
  philox_seed,
  philox_offset,
  causal_offset=_AUTO_OFFSET,
  rescale_threshold=-1.0,
  kv_bounds=None,
):
  B, Hq, Nq, D = q.shape
  o = q.new_empty((B, Hq, Nq, D))
  lse = q.new_empty((B, Hq, Nq), dtype=torch.float32)
  return o, lse


def ffpa_attn_forward_hip(
  q: torch.Tensor,
  k: torch.Tensor,
  v: torch.Tensor,
  attn_bias: torch.Tensor | None,
  *,
  causal: bool,
  softmax_scale: float,
  dropout_p: float = 0.0,
  philox_seed: int = 0,
  philox_offset: int = 0,
  causal_offset: int | None = None,
  rescale_threshold: float = -1.0,
  kv_bounds: torch.Tensor | None = None,
) -> tuple[torch.Tensor, torch.Tensor]:
  """Python-level entry (the analogue of ``_ffpa_attn_forward_cuda``, cuda/_ffpa_fwd.py:6-62):
  converts ``attn_bias=None`` into the empty tensor the op schema expects and calls the op.  ``kv_bounds``: precomputed key ranges of
  the mask (``mask_kv_bounds``), ``None`` = derived per call."""
  if attn_bias is None:
    attn_bias = q.new_empty((0,))
  return torch.ops.ffpa_attn._fwd_hip(
    q,
    k,
    v,
    attn_bias,
    0,
    1,
    int(causal),
    float(softmax_scale),
    float(dropout_p),
    int(philox_seed),
    int(philox_offset),
    _AUTO_OFFSET if causal_offset is None else int(causal_offset),
    float(rescale_threshold),
    kv_bounds,
  )


# ----------------------------------------------------------------------------------
# Packed sequences (ffpa_attn_varlen_func): the launch wrapper and its torch.library op.  Replaces the reference's
# ffpa_attn::_varlen_fwd_cute (src/ffpa_attn/cute/__init__.py:792-880), which only its CuTe-DSL backend serves.
# ----------------------------------------------------------------------------------
def _packed_rows(t: torch.Tensor) -> torch.Tensor:
  """[T, H, D] with head-dim stride 1, row / head strides multiples of 8 elements, non-overlapping rows and a 16-byte aligned base — else a copy."""
  ok = t.stride(-1) == 1 and t.stride(0) % 8 == 0 and t.stride(1) % 8 == 0 and t.data_ptr() % 16 == 0
  if ok and t.size(0) > 1 and t.stride(0) < t.size(2):
    ok = False
  return t if ok else t.contiguous()


def _varlen_params(q, k, v, o, lse, cu_seqlens_q, cu_seqlens_k, max_seqlen_q: int, max_seqlen_k: int, causal: bool, softmax_scale: float,
                   rescale_threshold: float, flags: int, seqused_k=None, num_splits: int = 0) -> FfpaVarlenFwdParams:
  p = FfpaVarlenFwdParams()
  p.struct_size = ctypes.sizeof(FfpaVarlenFwdParams)
  p.abi_version = ABI_VERSION
  p.q, p.k, p.v, p.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
  p.lse = lse.data_ptr() if lse is not None else None
  p.cu_seqlens_q, p.cu_seqlens_kv = cu_seqlens_q.data_ptr(), cu_seqlens_k.data_ptr()
  p.seqused_kv = seqused_k.data_ptr() if seqused_k is not None else None
  p.batch = cu_seqlens_q.numel() - 1
  p.heads_q, p.heads_kv, p.head_dim = q.size(1), k.size(1), q.size(2)
  p.max_seqlen_q, p.max_seqlen_kv = int(max_seqlen_q), int(max_seqlen_k)
  for name, t in (("q_stride", q), ("k_stride", k), ("v_stride", v), ("o_stride", o)):
    getattr(p, name)[:] = [t.stride(0), t.stride(1)]
  p.lse_stride_head = lse.stride(0) if lse is not None else 0
  p.dtype = _DTYPE[q.dtype]
  p.causal = 1 if causal else 0
  p.softmax_scale = float(softmax_scale)
  p.rescale_threshold = float(rescale_threshold)
  p.flags = int(flags) | (FLAG_DETERMINISTIC if os.environ.get("FFPA_HIP_DETERMINISTIC", "0").lower() not in ("0", "", "off", "false", "no") else 0)
  p.num_splits = int(num_splits)
  p.total_q = q.size(0)
  return p


# Scratch of a KV-split packed launch: a function of the shape class (ffpa_capi.hip varlen_plan), asked once per class
_VARLEN_SCRATCH: "dict[tuple, int]" = {}


def _varlen_scratch(lib, p: "FfpaVarlenFwdParams", device_index: int) -> int:
  if p.num_splits == 1 or p.flags & FLAG_DETERMINISTIC:
    return 0
  key = (id(lib), device_index, p.dtype, p.batch, p.heads_q, p.heads_kv, p.head_dim, p.max_seqlen_q, p.max_seqlen_kv, p.total_q, p.causal, p.flags, p.num_splits, os.environ.get("FFPA_HIP_FAKE_CUS"))
  hit = _VARLEN_SCRATCH.get(key)
  if hit is None:
    if len(_VARLEN_SCRATCH) >= 512:
      _VARLEN_SCRATCH.clear()
    hit = _VARLEN_SCRATCH[key] = int(lib.ffpa_attn_varlen_fwd_workspace_bytes(ctypes.byref(p)))
  return hit


def varlen_forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cu_seqlens_q: torch.Tensor, cu_seqlens_k: torch.Tensor, max_seqlen_q: int,
                   max_seqlen_k: int, causal: bool, softmax_scale: float, *, rescale_threshold: float = -1.0, return_lse: bool = True, flags: int = 0,
                   plan_out: "dict | None" = None, seqused_k: "torch.Tensor | None" = None, num_splits: int = 0):
  """One launch of the packed-sequence kernel: ``q [T_q, Hq, D]``, ``k`` / ``v [T_k, Hkv, D]``, int32 device ``cu_seqlens_*`` ``[B + 1]`` ->
  ``(o [T_q, Hq, D], lse [Hq, T_q] fp32 | None)``.  Nothing is read back to the host and nothing synchronises: the call captures into a HIP graph.
  Rows without a visible key: O = 0, LSE = -inf.

  ``seqused_k`` (int32 device ``[B]``, this build's extension at the op level — the public ``ffpa_attn_varlen_func`` rejects it like the reference's): sequence i
  uses only the first ``seqused_k[i]`` of its key rows — a KV cache of fixed capacity per sequence (``k`` / ``v`` = the cache viewed as ``[B * capacity, Hkv, D]``,
  ``cu_seqlens_k`` = multiples of the capacity) whose valid lengths live on the device: ONE captured graph serves every length, a replay follows lengths written
  in place, and with one token per sequence under GQA the group's heads are packed into the rows of one tile (FlashAttention's ``cache_seqlens`` decode).

  ``num_splits``: 0 = the library decides (launches that leave most of the chip idle — a decode batch of a few long sequences, a prefill chunk of one long
  prompt with a few heads per GPU — split every row tile's KV range over several workgroups and merge fp32 partials in a second kernel of the same call: equal
  to the unsplit launch to rounding, not to the bit), 1 = never, n = at most n.  ``FFPA_HIP_DETERMINISTIC=1`` / ``FLAG_DETERMINISTIC``: never."""
  if not q.is_cuda:
    raise NotImplementedError(f"ffpa_attn::_varlen_fwd_hip has no implementation for device '{q.device.type}' (the HIP kernel needs a GPU tensor)")
  lib = load_library()
  if q.dtype not in _DTYPE or k.dtype != q.dtype or v.dtype != q.dtype:
    raise TypeError(f"ffpa_attn::_varlen_fwd_hip only supports fp16/bf16 q/k/v of one dtype, got {q.dtype}, {k.dtype}, {v.dtype}")
  if q.dim() != 3 or k.dim() != 3 or v.dim() != 3:
    raise ValueError("ffpa_attn::_varlen_fwd_hip: q/k/v must be 3-D packed [T, H, D] tensors")
  if k.shape != v.shape or k.size(2) != q.size(2):
    raise ValueError(f"ffpa_attn::_varlen_fwd_hip: k {tuple(k.shape)} and v {tuple(v.shape)} must share their shape and q's head dim ({q.size(2)})")
  if k.size(1) == 0 or q.size(1) % k.size(1) != 0:
    raise ValueError(f"ffpa_attn::_varlen_fwd_hip: query num_heads ({q.size(1)}) must be a multiple of key/value num_heads ({k.size(1)})")
  for name, cu in (("cu_seqlens_q", cu_seqlens_q), ("cu_seqlens_k", cu_seqlens_k)):
    if cu.dtype != torch.int32 or cu.dim() != 1 or cu.numel() < 2:
      raise ValueError(f"ffpa_attn::_varlen_fwd_hip: {name} must be a 1-D int32 tensor of length batch + 1")
    if cu.device != q.device:
      raise ValueError(f"ffpa_attn::_varlen_fwd_hip: {name} must be on q's device, got {cu.device} and {q.device}")
  if cu_seqlens_q.numel() != cu_seqlens_k.numel():
    raise ValueError("ffpa_attn::_varlen_fwd_hip: cu_seqlens_q and cu_seqlens_k must have one length")
  if seqused_k is not None:
    if seqused_k.dtype != torch.int32 or seqused_k.dim() != 1 or seqused_k.numel() != cu_seqlens_k.numel() - 1 or seqused_k.device != q.device:
      raise ValueError("ffpa_attn::_varlen_fwd_hip: seqused_k must be a 1-D int32 tensor of length batch on q's device")
    seqused_k = seqused_k.contiguous()
  if k.device != q.device or v.device != q.device:
    raise ValueError(f"ffpa_attn::_varlen_fwd_hip: q/k/v must be on one device, got {q.device}, {k.device}, {v.device}")
  Tq, Hq, D = q.shape
  Dp = (D + 7) // 8 * 8  # rows must be whole 16-byte slots: only a head dim that is not a multiple of 8 is padded (copies)
  if Dp != D:
    pad = (0, Dp - D)
    q, k, v = (torch.nn.functional.pad(t, pad) for t in (q, k, v))
  if k.size(0) == 0:
    # no key row anywhere: every output row is the empty row (O = 0, LSE = -inf).  The C-ABI wants non-NULL bases; nothing is read through them
    k = v = q.new_zeros((1, k.size(1), Dp))
  q, k, v = _packed_rows(q), _packed_rows(k), _packed_rows(v)
  cu_seqlens_q, cu_seqlens_k = cu_seqlens_q.contiguous(), cu_seqlens_k.contiguous()
  o = torch.empty((Tq, Hq, Dp), dtype=q.dtype, device=q.device)
  lse = torch.empty((Hq, Tq), dtype=torch.float32, device=q.device) if return_lse else None
  if Tq == 0 or max_seqlen_q <= 0:
    return (o[..., :D] if Dp != D else o), lse  # (nothing to compute: no query row in any sequence)
  p = _varlen_params(q, k, v, o, lse, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal, softmax_scale, rescale_threshold, flags, seqused_k, num_splits)
  with torch.cuda.device(q.device):
    stream = torch.cuda.current_stream(q.device).cuda_stream
    ws_bytes = _varlen_scratch(lib, p, q.device.index or 0)
    workspace = None
    if ws_bytes:
      workspace = _workspace(q.device, stream, ws_bytes)  # (held in a local until the launch below has been enqueued)
      p.workspace = workspace.data_ptr()
      p.workspace_bytes = ws_bytes
    if plan_out is not None:
      plan = (ctypes.c_int * 5)()
      if lib.ffpa_attn_varlen_fwd_plan(ctypes.byref(p), plan) == 0:
        plan_out.update(row_tiles=plan[0], block_rows=plan[1], block_keys=plan[2], workgroups=plan[3], splits=plan[4])
      name = ctypes.create_string_buffer(200)
      if lib.ffpa_attn_varlen_fwd_kernel(ctypes.byref(p), name, len(name)) == 0:
        plan_out["kernel"] = name.value.decode()
    rc = lib.ffpa_attn_varlen_fwd(ctypes.byref(p), ctypes.c_void_p(stream))
  if rc != 0:
    raise _STATUS_EXC.get(rc, RuntimeError)(f"ffpa_attn_varlen_fwd: {lib.ffpa_attn_last_error().decode()} (status {rc})")
  if Dp != D:
    o = o[..., :D].contiguous()
  return o, lse


def varlen_launch_plan(batch: int, heads_q: int, heads_kv: int, max_seqlen_q: int, max_seqlen_k: int, head_dim: int, *,
                       dtype: torch.dtype = torch.bfloat16, causal: bool = False, flags: int = 0, total_q: int = 0, num_splits: int = 0) -> dict:
  """The packed-sequence launch for a shape class, without launching (placeholder pointers): row tiles per (sequence, head), tile, workgroups, KV ranges per
  sequence, kernel name.  ``total_q`` (rows of q) > 0: the plan of a call that hands the library its scratch (``varlen_forward`` does) — KV splits included;
  0: the plan without scratch (never split)."""
  lib = load_library()
  d8 = (int(head_dim) + 7) // 8 * 8
  p = FfpaVarlenFwdParams()
  p.struct_size = ctypes.sizeof(FfpaVarlenFwdParams)
  p.abi_version = ABI_VERSION
  p.q = p.k = p.v = p.o = p.cu_seqlens_q = p.cu_seqlens_kv = 16
  p.batch, p.heads_q, p.heads_kv, p.head_dim = int(batch), int(heads_q), int(heads_kv), d8
  p.max_seqlen_q, p.max_seqlen_kv = int(max_seqlen_q), int(max_seqlen_k)
  for name, h in (("q_stride", heads_q), ("k_stride", heads_kv), ("v_stride", heads_kv), ("o_stride", heads_q)):
    getattr(p, name)[:] = [h * d8, d8]
  p.dtype = _DTYPE[dtype]
  p.causal = 1 if causal else 0
  p.softmax_scale = float(head_dim) ** -0.5
  p.rescale_threshold = -1.0
  p.flags = int(flags)
  p.num_splits = int(num_splits)
  p.total_q = int(total_q)
  if total_q > 0:
    p.workspace, p.workspace_bytes = 16, 0xFFFFFFFFFFFFFFFF
  plan = (ctypes.c_int * 5)()
  rc = lib.ffpa_attn_varlen_fwd_plan(ctypes.byref(p), plan)
  if rc != 0:
    raise _STATUS_EXC.get(rc, RuntimeError)(lib.ffpa_attn_last_error().decode())
  name = ctypes.create_string_buffer(200)
  lib.ffpa_attn_varlen_fwd_kernel(ctypes.byref(p), name, len(name))
  out = {"row_tiles": plan[0], "block_rows": plan[1], "block_keys": plan[2], "workgroups": plan[3], "kernel": name.value.decode()}
  if total_q > 0:
    out["splits"] = plan[4]
  return out


torch.library.define(
  f"{_OP_NAMESPACE}::_varlen_fwd_hip",
  "(Tensor q, Tensor k, Tensor v, Tensor cu_seqlens_q, Tensor cu_seqlens_k, int max_seqlen_q, int max_seqlen_k, "
  "float softmax_scale, int causal, float rescale_threshold=-1.0, Tensor? seqused_k=None) -> (Tensor o, Tensor softmax_lse)",
)


@torch.library.impl(f"{_OP_NAMESPACE}::_varlen_fwd_hip", "CUDA")  # ROCm tensors dispatch on the CUDA key
def _varlen_fwd_hip_torch_op(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, softmax_scale, causal, rescale_threshold=-1.0, seqused_k=None):
  return varlen_forward(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, bool(causal), softmax_scale,
                        rescale_threshold=rescale_threshold, return_lse=True, seqused_k=seqused_k)


@torch.library.register_fake(f"{_OP_NAMESPACE}::_varlen_fwd_hip")
def _varlen_fwd_hip_fake(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, softmax_scale, causal, rescale_threshold=-1.0, seqused_k=None):
  total_q, heads, head_dim = q.shape
  return q.new_empty((total_q, heads, head_dim)), q.new_empty((heads, total_q), dtype=torch.float32)
