"""Public entry point: ``ffpa_attn_func`` with the ``scaled_dot_product_attention`` signature.

Drop-in for the reference's ``ffpa_attn_func`` (``src/ffpa_attn/ffpa_attn_interface.py:71-189``):
same positional/keyword arguments, same SDPA short-circuit for what the fused kernel does not
serve (small D, D > 1024, short sequences; ``functional.py:676-724``), same validation errors,
and — for everything else — the hand-written gfx950 kernel behind ``ffpa_attn::_fwd_hip``.
"""

from __future__ import annotations

import torch

from .functional import FFPAAttnMeta


def _reserve_dropout_rng(query: torch.Tensor, key: torch.Tensor, dropout_p: float) -> tuple[int, int]:
  """(seed, offset) for this call's dropout, advancing the device generator by one Philox output per logical
  score, rounded up to a multiple of four — what SDPA's efficient attention reserves and what the reference
  does (functional.py:518-540)."""
  if dropout_p <= 0.0:
    return 0, 0
  with torch.cuda.device(query.device):
    seed = int(torch.cuda.initial_seed())
    offset = int(torch.cuda._get_rng_state_offset())
    elems = query.size(0) * query.size(1) * query.size(2) * key.size(2)
    torch.cuda._set_rng_state_offset(offset + (elems + 3) // 4 * 4)
  return seed, offset


class _FFPAAttnFunc(torch.autograd.Function):
  """HIP forward + SDPA-style backward (the reference's ``_FFPAAttnFunc`` with ``forward_backend`` =
  native kernel and ``backward_backend="sdpa"``, functional.py:964-1172).  O and the natural-log LSE
  ``[B, Hq, Nq]`` are saved only when a gradient is needed (functional.py:1066-1077)."""

  @staticmethod
  def forward(ctx, query, key, value, attn_bias, meta: FFPAAttnMeta):
    from .hip import ffpa_attn_forward_hip

    thr = getattr(meta.forward_meta, "rescale_threshold", None)
    seed, offset = _reserve_dropout_rng(query, key, meta.attn_meta.dropout_p)
    out, lse = ffpa_attn_forward_hip(
      query,
      key,
      value,
      attn_bias,
      causal=meta.attn_meta.is_causal,
      softmax_scale=meta.attn_meta.scale,
      dropout_p=meta.attn_meta.dropout_p,
      philox_seed=seed,
      philox_offset=offset,
      rescale_threshold=-1.0 if thr is None else float(thr),
      kv_bounds=getattr(meta.forward_meta, "kv_bounds", None) if attn_bias is not None else None,
    )
    needs_grad = meta.attn_meta.is_grad_enabled and any(
      t is not None and t.requires_grad for t in (query, key, value, attn_bias)
    )
    if needs_grad:
      ctx.save_for_backward(query, key, value, out, lse, attn_bias)
      ctx.causal = meta.attn_meta.is_causal
      ctx.scale = meta.attn_meta.scale
      ctx.dropout = (meta.attn_meta.dropout_p, seed, offset) if meta.attn_meta.dropout_p > 0.0 else None
    return out

  @staticmethod
  def backward(ctx, grad_out):
    from .backward import attention_backward

    query, key, value, out, lse, attn_bias = ctx.saved_tensors
    want_bias = attn_bias is not None and ctx.needs_input_grad[3]
    dq, dk, dv, dbias = attention_backward(
      grad_out.contiguous(), query, key, value, out, lse, causal=ctx.causal, scale=ctx.scale, attn_bias=attn_bias,
      want_bias_grad=want_bias, dropout=ctx.dropout,
    )
    return dq, dk, dv, dbias, None


def _plain_call(query, key, value, attn_bias) -> bool:
  """May this call skip the dispatcher?  Only when nobody is listening on it and every tensor is a plain ``torch.Tensor``: fake / functional /
  subclass tensors (the mask included — it reaches ctypes as a raw pointer), an active ``TorchDispatchMode`` (FlopCounterMode, profiler op records,
  FakeTensorMode) and functorch transforms (vmap's BatchedTensor has type ``torch.Tensor``) all go through the registered op
  ``ffpa_attn::_fwd_hip``, which has the rules for them."""
  if not (type(query) is torch.Tensor and type(key) is torch.Tensor and type(value) is torch.Tensor):
    return False
  if attn_bias is not None and type(attn_bias) is not torch.Tensor:
    return False
  if torch._C._len_torch_dispatch_stack() != 0:
    return False
  if torch._C._functorch.peek_interpreter_stack() is not None:
    return False
  return True


@torch._dynamo.disable
def _ffpa_apply(query, key, value, attn_bias, meta: FFPAAttnMeta) -> torch.Tensor:
  """Graph-break boundary, as the reference's ``_ffpa_apply`` (functional.py:1195-1216).  A call nothing will ever differentiate —
  inference: grad mode off, or no input requires a gradient — goes straight to the launch wrapper: no autograd node, no dispatcher
  round trip, no LSE tensor (what a decode step that synchronises per token pays on the host)."""
  needs_grad = meta.attn_meta.is_grad_enabled and any(t is not None and t.requires_grad for t in (query, key, value, attn_bias))
  if not needs_grad and query.is_cuda and _plain_call(query, key, value, attn_bias):
    from . import hip

    thr = getattr(meta.forward_meta, "rescale_threshold", None)
    seed, offset = _reserve_dropout_rng(query, key, meta.attn_meta.dropout_p)
    return hip.forward(query, key, value, attn_bias, meta.attn_meta.is_causal, meta.attn_meta.scale, dropout_p=meta.attn_meta.dropout_p,
                       philox_seed=seed, philox_offset=offset, rescale_threshold=-1.0 if thr is None else float(thr), return_lse=False,
                       kv_bounds=getattr(meta.forward_meta, "kv_bounds", None) if attn_bias is not None else None)[0]
  return _FFPAAttnFunc.apply(query, key, value, attn_bias, meta)


def ffpa_attn_func(
  query: torch.Tensor,
  key: torch.Tensor,
  value: torch.Tensor,
  attn_mask: torch.Tensor | None = None,
  dropout_p: float = 0.0,
  is_causal: bool = False,
  scale: float | None = None,
  enable_gqa: bool = False,
  **kwargs: object,
) -> torch.Tensor:
  """Fused attention forward for large head dims (``256 < D <= 1024``) on MI355X.

  ``query`` is ``[B, Hq, Nq, D]``; ``key`` / ``value`` are ``[B, Hkv, Nkv, D]`` (``Hq % Hkv == 0``,
  pass ``enable_gqa=True`` when they differ); fp16 or bf16.  ``attn_mask`` follows SDPA
  (bool = keep, float = additive, broadcastable to ``[B, Hq, Nq, Nkv]``) and excludes
  ``is_causal``.  ``is_causal=True`` masks ``key > row + (Nkv - Nq)`` (queries aligned to the KV
  tail — FlashAttention's convention, NOT SDPA's top-left one) and requires ``Nkv >= Nq``.
  ``scale`` defaults to ``1/sqrt(D)``.

  Masks: the kernel clips its KV walk to the key ranges of ``attn_mask`` and skips the mask reads where the mask is neutral (an explicit causal
  mask then costs what ``is_causal`` costs).  The ranges come from a scan kernel — a full read of the mask — that runs on EVERY call with a mask
  worth scanning; a serving loop with a static mask should scan once (``hip.mask_kv_bounds(mask, Nq, Nkv)``) and pass the result as
  ``forward_backend=HIPBackend(forward=True, kv_bounds=ranges)`` (or ``TritonBackend`` / ``CUDABackend``: the field exists on all three), or opt
  into the per-tensor cache with ``FFPA_HIP_MASK_BOUNDS_CACHE=1`` (safe only for masks written through torch ops alone: ``hip.forward``).

  Extra keywords: ``backend``, ``forward_backend``, ``backward_backend`` — a name
  (``"hip"``, ``"cuda"``, ``"triton"``, ``"cutedsl"``, ``"sdpa"``) or a
  :class:`~ffpa_attn_amd.backends.Backend` instance.  Anything else raises ``TypeError``.

  Calls the kernel does not serve go to ``torch._C._nn.scaled_dot_product_attention`` directly
  (never through ``F.scaled_dot_product_attention``, so monkey-patching that symbol with this
  function cannot recurse — tests/test_monkey_patch.py:65-69 in the reference).
  """
  meta = FFPAAttnMeta.from_kwargs(**kwargs)
  if meta.fallback(query, key, attn_mask, dropout_p, is_causal=is_causal):
    return torch._C._nn.scaled_dot_product_attention(
      query,
      key,
      value,
      attn_mask=attn_mask,
      dropout_p=dropout_p,
      is_causal=is_causal,
      scale=scale,
      enable_gqa=enable_gqa,
    )
  meta, query, key, value, attn_bias = meta.normalize(
    query, key, value, attn_mask, dropout_p, is_causal, scale, enable_gqa
  )
  return _ffpa_apply(query, key, value, attn_bias, meta)


def ffpa_attn_varlen_func(
  q: torch.Tensor,
  k: torch.Tensor,
  v: torch.Tensor,
  cu_seqlens_q: torch.Tensor,
  cu_seqlens_k: torch.Tensor | None,
  max_seqlen_q: int,
  max_seqlen_k: int,
  *,
  dropout_p: float = 0.0,
  softmax_scale: float | None = None,
  causal: bool = False,
  enable_gqa: bool = False,
  return_lse: bool = False,
  **kwargs: object,
):
  """FFPA variable-length attention over packed sequences (FlashAttention's THD layout) — the reference's entry point under its name and signature
  (``src/ffpa_attn/ffpa_attn_interface.py:192-279``).  ``q`` is ``[T_q, H_q, D]``, ``k`` / ``v`` are ``[T_k, H_kv, D]``; ``cu_seqlens_q`` /
  ``cu_seqlens_k`` (int32 device tensors of length ``B + 1`` starting at 0; ``cu_seqlens_k=None`` = self-attention) mark the sequences.

  In the reference the CuTe-DSL backend ALONE serves this call (NVIDIA SM8x / SM90 / SM100; head dims 320 ... 1024).  Here it runs on the MI355X
  kernel of the dense path — ONE launch for the whole batch, the boundaries read on the device: nothing is copied to the host, the call never
  synchronises (the reference reads ``cu_seqlens`` back to check ``max_seqlen``) and captures into a HIP graph.  Same checks, same exception classes and
  texts (``varlen.py``); head dims: any up to 1024.

  ``causal`` = the reference's lower-right (tail-aligned) mask per sequence.  Rows without a visible key — an empty key sequence; the first
  ``N_q - N_k`` rows of a causal sequence with more queries than keys — come out as ``O = 0`` and ``LSE = -inf``
  (``tests/test_ffpa_cute_sm100.py:1117-1183``).  ``max_seqlen_q`` must be >= the longest query sequence (rows past it are not computed).
  ``dropout_p`` must be 0 and every FlashAttention-varlen extension (``window_size``, ``softcap``, ``seqused_k``, ``block_table`` ...) raises
  ``NotImplementedError``, as in the reference.  Returns ``out [T_q, H_q, D]`` — and ``lse [H_q, T_q]`` fp32 with ``return_lse=True``."""
  from .varlen import varlen_apply

  return varlen_apply(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=dropout_p, softmax_scale=softmax_scale,
                      causal=causal, enable_gqa=enable_gqa, return_lse=return_lse, kwargs=dict(kwargs))
