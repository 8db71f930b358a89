"""Public entry point: ``ffpa_attn_func`` with the ``scaled_dot_product_attention`` signature.

Drop-in for the reference's ``ffpa_attn_func`` (``src/ffpa_attn/ffpa_attn_interface.py:71-189``):
same positional/keyword arguments, same SDPA short-circuit for what the fused kernel does not
serve (small D, D > 1024, short sequences; ``functional.py:676-724``), same validation errors,
and — for everything else — the hand-written gfx950 kernel behind ``ffpa_attn::_fwd_hip``.
"""

from __future__ import annotations

import torch

from .functional import FFPAAttnMeta


@torch._dynamo.disable
def _ffpa_apply(query, key, value, attn_bias, meta: FFPAAttnMeta) -> torch.Tensor:
  """Graph-break boundary, as the reference's ``_ffpa_apply`` (functional.py:1195-1216)."""
  from .hip import ffpa_attn_forward_hip

  fwd = meta.forward_meta
  thr = getattr(fwd, "rescale_threshold", None)
  out, _lse = ffpa_attn_forward_hip(
    query,
    key,
    value,
    attn_bias,
    causal=meta.attn_meta.is_causal,
    softmax_scale=meta.attn_meta.scale,
    dropout_p=meta.attn_meta.dropout_p,
    rescale_threshold=-1.0 if thr is None else float(thr),
  )
  return out


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

  Extra keywords: ``backend``, ``forward_backend``, ``backward_backend`` — a name
  (``"hip"``, ``"cuda"``, ``"triton"``, ``"cutedsl"``, ``"sdpa"``) or a
  :class:`~ffpa_attn_amd.backends.Backend` instance.  Anything else raises ``TypeError``.

  Calls the kernel does not serve go to ``torch._C._nn.scaled_dot_product_attention`` directly
  (never through ``F.scaled_dot_product_attention``, so monkey-patching that symbol with this
  function cannot recurse — tests/test_monkey_patch.py:65-69 in the reference).
  """
  meta = FFPAAttnMeta.from_kwargs(**kwargs)
  if meta.fallback(query, key, attn_mask, dropout_p):
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
