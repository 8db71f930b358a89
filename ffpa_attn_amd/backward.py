"""Backward for the large-D path, driven by the forward kernel's O and LSE.

The north star is forward-only; this is SURVEY.md §8f rank 2 ("LSE contract + SDPA backward hookup"):
the cheapest route to training usability without a backward kernel.  Same plan as the reference's
``backward_backend="sdpa"`` (``src/ffpa_attn/aten/_efficient_bwd.py:50-186``): hand PyTorch's
efficient-attention backward the saved ``q, k, v, O, LSE`` — expanding K/V for GQA and reducing their
gradients back, turning a tail-aligned causal mask with ``Nq != Nkv`` into an explicit bias (the aten op's
``is_causal`` is top-left aligned), padding LSE rows to a multiple of 8.  If the aten op rejects the
shape on this ROCm build, a row-chunked recomputation in plain torch ops produces the same gradients.
"""

from __future__ import annotations

import torch

from .functional import warning_once


def _tail_aligned_bias(q, k, dtype) -> torch.Tensor:
  nq, nkv = q.size(2), k.size(2)
  rows = torch.arange(nq, device=q.device).view(-1, 1)
  cols = torch.arange(nkv, device=q.device).view(1, -1)
  keep = cols <= rows + (nkv - nq)
  return torch.zeros(1, 1, nq, nkv, dtype=dtype, device=q.device).masked_fill_(~keep, float("-inf"))


def _reduce_groups(g: torch.Tensor, like: torch.Tensor, group: int) -> torch.Tensor:
  if group == 1:
    return g.to(like.dtype)
  B, Hkv, N, D = like.shape
  return g.reshape(B, Hkv, group, N, D).sum(dim=2).to(like.dtype)


def _aten_efficient_backward(grad_out, q, k, v, o, lse, causal, scale, attn_bias, want_bias_grad):
  group = q.size(1) // k.size(1)
  causal_for_op = causal
  bias = attn_bias
  if causal and q.size(2) != k.size(2):
    bias = _tail_aligned_bias(q, k, q.dtype)
    causal_for_op = False
  if bias is not None:
    bias = bias.to(q.dtype).expand(q.size(0), q.size(1), q.size(2), k.size(2))
  o = o.transpose(1, 2).contiguous().transpose(1, 2)
  if lse.size(1) > 1 and lse.stride(1) % 8 != 0:
    padded = lse.new_empty(*lse.shape[:-1], (lse.size(-1) + 7) // 8 * 8)
    padded[..., : lse.size(-1)] = lse
    lse = padded
  kx = k.repeat_interleave(group, dim=1).contiguous() if group > 1 else k
  vx = v.repeat_interleave(group, dim=1).contiguous() if group > 1 else v
  zero = torch.zeros(1, dtype=torch.int64)
  dq, dk, dv, dbias = torch.ops.aten._scaled_dot_product_efficient_attention_backward.default(
    grad_out, q, kx, vx, bias, o, lse, zero, zero, 0.0,
    (True, True, True, bool(want_bias_grad and attn_bias is not None)), causal_for_op, scale=scale,
  )
  dk, dv = _reduce_groups(dk, k, group), _reduce_groups(dv, v, group)
  if want_bias_grad and attn_bias is not None:
    dbias = dbias.sum_to_size(attn_bias.shape).to(attn_bias.dtype)
  else:
    dbias = None
  return dq.to(q.dtype), dk, dv, dbias


def _chunked_recompute_backward(grad_out, q, k, v, o, lse, causal, scale, attn_bias, want_bias_grad,
                                budget_bytes: int = 1 << 30, dropout=None):
  """dV = P^T dO;  dS = P o (dO V^T - rowsum(dO o O));  dQ = scale dS K;  dK = scale dS^T Q, with
  P = exp(scale QK^T + bias - LSE) recomputed per block of query rows in fp32."""
  B, Hq, Nq, D = q.shape
  Hkv, Nkv = k.size(1), k.size(2)
  group = Hq // Hkv
  kx = (k.repeat_interleave(group, dim=1) if group > 1 else k).float()
  vx = (v.repeat_interleave(group, dim=1) if group > 1 else v).float()
  go = grad_out.float()
  delta = (go * o.float()).sum(-1)
  dq = torch.empty_like(q, dtype=torch.float32)
  dk = torch.zeros_like(kx)
  dv = torch.zeros_like(vx)
  dbias_full = None
  if want_bias_grad and attn_bias is not None:
    dbias_full = torch.zeros(B, Hq, Nq, Nkv, dtype=torch.float32, device=q.device)
  chunk = max(16, min(Nq, budget_bytes // max(1, B * Hq * Nkv * 4 * (3 if dropout is None else 12))))
  if dropout is not None:
    from .philox import dropout_keep_mask

    p_drop, seed, offset = dropout
    keep_scale = 1.0 / (1.0 - p_drop)
    bh = (torch.arange(B, device=q.device).view(B, 1, 1, 1) * Hq + torch.arange(Hq, device=q.device).view(1, Hq, 1, 1))
  cols = torch.arange(Nkv, device=q.device).view(1, -1)
  for r0 in range(0, Nq, chunk):
    r1 = min(Nq, r0 + chunk)
    qc = q[:, :, r0:r1].float()
    s = (qc @ kx.transpose(-1, -2)) * scale
    if attn_bias is not None:
      bc = attn_bias if attn_bias.size(2) == 1 else attn_bias[:, :, r0:r1]
      s = s + bc.float()
    if causal:
      rows = torch.arange(r0, r1, device=q.device).view(-1, 1)
      s = s.masked_fill(cols > rows + (Nkv - Nq), float("-inf"))
    p = torch.exp(s - lse[:, :, r0:r1, None])
    p = torch.nan_to_num(p, nan=0.0)  # fully masked rows: LSE = -inf
    goc = go[:, :, r0:r1]
    if dropout is None:
      dv += p.transpose(-1, -2) @ goc
      dp = goc @ vx.transpose(-1, -2)
    else:
      # forward: O = (keep o round(P) / (1-p)) V / l with the row sum undropped; d/dP of that
      rows_i = torch.arange(r0, r1, device=q.device).view(1, 1, -1, 1)
      idx = (bh * Nq + rows_i) * Nkv + cols.view(1, 1, 1, -1)
      keep = dropout_keep_mask(seed, offset, idx, p_drop).to(torch.float32) * keep_scale
      dv += (p * keep).transpose(-1, -2) @ goc
      dp = (goc @ vx.transpose(-1, -2)) * keep
    ds = p * (dp - delta[:, :, r0:r1, None])
    dq[:, :, r0:r1] = (ds @ kx) * scale
    dk += (ds.transpose(-1, -2) @ qc) * scale
    if dbias_full is not None:
      dbias_full[:, :, r0:r1] = ds
  dbias = None
  if dbias_full is not None:
    dbias = dbias_full.sum_to_size(attn_bias.shape).to(attn_bias.dtype)
  return dq.to(q.dtype), _reduce_groups(dk, k, group), _reduce_groups(dv, v, group), dbias


# (dtype, head_dim) -> False once the aten op has said it cannot serve that head dim / dtype at all.  Only such
# capability refusals are remembered; a failure that may be specific to one call (bias layout, alignment, out of
# memory) falls back for that call only and aten is tried again next time.
_aten_unsupported: dict = {}
# (generic words such as "not supported" / "unsupported" are NOT markers: aten uses them for per-call refusals too — a bias stride, an
# alignment — and remembering one of those would send every later backward of that (dtype, D) down the slower recompute path)
# Bare "dtype" / "data type" are not markers either (round-3 advisor finding): aten words per-call refusals of a BIAS dtype that way; a dtype
# message counts only when it names query / key / value.
_CAPABILITY_MARKERS = ("head_dim", "head dim", "headdim", "head size", "last dimension", "no available kernel", "no kernel")
_DTYPE_WORDS = ("dtype", "data type")
_QKV_WORDS = ("query", "key", "value")


def _is_capability_error(exc: Exception) -> bool:
  if isinstance(exc, torch.OutOfMemoryError):
    return False
  msg = str(exc).lower()
  if "out of memory" in msg:
    return False
  if isinstance(exc, NotImplementedError) or any(m in msg for m in _CAPABILITY_MARKERS):
    return True
  return any(w in msg for w in _DTYPE_WORDS) and any(w in msg for w in _QKV_WORDS) and "bias" not in msg and "mask" not in msg


def _additive_bias(attn_bias, dtype):
  """The additive form of a boolean mask (0 / -inf in ``dtype``, what the reference builds for every call,
  functional.py:891-898): only the backward implementations need it."""
  if attn_bias is not None and attn_bias.dtype in (torch.bool, torch.uint8):
    return torch.zeros_like(attn_bias, dtype=dtype).masked_fill_(attn_bias == 0, float("-inf"))
  return attn_bias


def attention_backward(grad_out, q, k, v, o, lse, *, causal: bool, scale: float, attn_bias=None,
                       want_bias_grad: bool = False, force: str | None = None, dropout=None):
  """``(dq, dk, dv, d_attn_bias)`` for the forward ``o, lse = ffpa(q, k, v)``.  ``force`` = ``"aten"`` /
  ``"recompute"`` pins one implementation (tests); otherwise aten first, recompute if it refuses."""
  key = (q.dtype, q.size(-1))
  attn_bias = _additive_bias(attn_bias, q.dtype)
  if dropout is not None:
    # the fused aten backward would regenerate a DIFFERENT mask on ROCm: rebuild the kernel's own (philox.py)
    return _chunked_recompute_backward(grad_out, q, k, v, o, lse, causal, scale, attn_bias, want_bias_grad, dropout=dropout)
  if force != "recompute" and not _aten_unsupported.get(key, False):
    try:
      return _aten_efficient_backward(grad_out, q, k, v, o, lse, causal, scale, attn_bias, want_bias_grad)
    except (RuntimeError, NotImplementedError) as e:
      if force == "aten":
        raise
      if isinstance(e, torch.OutOfMemoryError) or "out of memory" in str(e).lower():
        raise  # the recompute path needs more memory, not less
      if _is_capability_error(e):
        _aten_unsupported[key] = True
      warning_once(f"ffpa_attn_func: aten efficient-attention backward refused D={q.size(-1)} "
                   f"({str(e).splitlines()[0][:120]}); using the chunked recompute backward"
                   + (" from now on" if _aten_unsupported.get(key) else " for this call"))
  return _chunked_recompute_backward(grad_out, q, k, v, o, lse, causal, scale, attn_bias, want_bias_grad)
