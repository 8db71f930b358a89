"""Boolean masks on the kernel's own path (FFPA_BIAS_BOOL8) and the boundary hygiene of C-ABI v3 (`pytest -m gpu`).

The reference turns every boolean ``attn_mask`` into a 0 / -inf tensor in q's dtype on the host before the launch
(src/ffpa_attn/functional.py:891-898) and adds it as a bias (native/prefill.cuh:556-658).  Here the kernel reads the
caller's bytes: ``score + 0`` / ``score + (-inf)`` and ``keep score`` / ``-inf`` are the same numbers, so every result
must be BIT-identical to the additive form — that is what these tests pin, next to SDPA and the oracle.
"""

import ctypes

import pytest
import torch
import torch.nn.functional as F

from test_fwd_gpu import _check_vs_oracle, _close, _f32, _rand, _within_north_star, hip  # noqa: F401  (fixture + helpers)

pytestmark = pytest.mark.gpu


def _additive(mask, dtype):
  return torch.zeros(mask.shape, dtype=dtype, device=mask.device).masked_fill(~mask, float("-inf"))


def _same_bits(a, b):
  return torch.equal(torch.nan_to_num(a.float(), nan=7.0), torch.nan_to_num(b.float(), nan=7.0))


@pytest.mark.parametrize("shape", [(1, 1, 1, 600), (2, 1, 1, 600), (1, 4, 1, 600), (1, 1, 520, 600), (2, 4, 520, 600), (2, 1, 520, 1),
                                   (1, 4, 520, 600)])
@pytest.mark.parametrize("D", [320, 512, 1024])
def test_bool_mask_is_bit_identical_to_its_additive_form(hip, shape, D):
  B, H, Nq, Nkv = 2, 4, 520, 600  # Nkv = 600: the last tile is ragged (byte loads), 600 % 16 != 0 rows are unaligned
  q, k, v = _rand((B, H, Nq, D), seed=51), _rand((B, H, Nkv, D), seed=52), _rand((B, H, Nkv, D), seed=53)
  g = torch.Generator(device="cuda").manual_seed(sum(shape) * 7 + D)
  mask = torch.rand(shape, device="cuda", generator=g) > 0.25
  mask[..., 0] = True
  # the boolean build sets the masked scores to -inf after the MFMA chain, the additive build starts the chain from 0 / -inf (bias / scale):
  # the same scores, bit for bit — where both builds walk the same tiles (the additive build has 64-key tiles at every head dim <= 512,
  # the boolean one 128 keys at D = 320: another order of the online softmax, equal up to rounding there)
  ob, lb = hip.forward(q, k, v, mask, False, D ** -0.5, kv_bounds=False)
  for flags in (0, hip.FLAG_NO_BIAS_LDS):  # the additive form staged through LDS / read element by element
    oa, la = hip.forward(q, k, v, _additive(mask, q.dtype), False, D ** -0.5, kv_bounds=False, flags=flags)
    if D == 320:
      _close(oa, ob, q.dtype, f"{shape} D={D}")
      assert (la - lb).abs().max().item() <= 2e-5
    else:
      assert torch.equal(ob, oa) and torch.equal(lb, la), f"{shape} D={D} flags={flags}"
  if shape[3] == 1:
    # a [B, 1, Nq, 1] mask (no key axis): PyTorch-ROCm's fused SDPA is not a usable reference for it (profiles/NOTES.md section 4: it moves the
    # result for the additive form of such a mask, and for the boolean form it returned NaN rows under an all-True mask on this pool) —
    # fp32 math instead
    s_ = (q.float() @ k.float().transpose(-1, -2)) * D ** -0.5
    want = torch.softmax(s_.masked_fill(~mask, float("-inf")), -1) @ v.float()
    _close(ob, want.to(q.dtype), q.dtype, f"math {shape}")
  else:
    _close(ob, F.scaled_dot_product_attention(q, k, v, attn_mask=mask), q.dtype, f"sdpa {shape}")
  if D == 320:
    _check_vs_oracle(ob, lb, q, k, v, bias=_f32(_additive(mask, torch.float32)), name=f"bool {shape}")


def test_bool_mask_vector_and_byte_paths_tails_and_nan_rows(hip):
  """16-byte mask loads need unit key stride and 16-byte aligned rows; everything else walks bytes.  Fully masked rows
  give NaN (SDPA semantics), a row whose first tiles are hidden stays finite."""
  D = 512
  for Nq, Nkv in ((513, 1024), (513, 1000), (130, 777)):  # aligned rows / ragged last tile / unaligned rows
    q, k, v = _rand((1, 2, Nq, D), seed=61), _rand((1, 2, Nkv, D), seed=62), _rand((1, 2, Nkv, D), seed=63)
    mask = torch.rand(1, 2, Nq, Nkv, device="cuda") > 0.3
    mask[0, 0, 5, :] = False
    mask[0, :, 9, :128] = False
    mask[0, 1, 100, 1:] = False
    mask[0, 1, 100, 0] = True
    ob, lb = hip.forward(q, k, v, mask, False, D ** -0.5, kv_bounds=False)
    oa, la = hip.forward(q, k, v, _additive(mask, q.dtype), False, D ** -0.5, kv_bounds=False, flags=hip.FLAG_NO_BIAS_LDS)
    assert _same_bits(ob, oa) and _same_bits(lb, la), (Nq, Nkv)
    assert torch.isnan(ob[0, 0, 5]).all() and torch.isfinite(ob[:, :, 9]).all()
    assert torch.equal(ob[0, 1, 100], v[0, 1, 0])
    # a strided view (key stride 2) takes the byte path; a uint8 tensor is NOT a mask (the reference accepts bool / float masks only)
    wide = torch.zeros(1, 2, Nq, 2 * Nkv, dtype=torch.bool, device="cuda")
    wide[..., ::2] = mask
    os_, _ = hip.forward(q, k, v, wide[..., ::2], False, D ** -0.5, kv_bounds=False)
    assert _same_bits(os_, oa)
    with pytest.raises(TypeError):
      hip.forward(q, k, v, mask.to(torch.uint8), False, D ** -0.5, kv_bounds=False)


def test_bool_mask_bounds_match_the_additive_scan_and_clip_the_same_tiles(hip):
  B, Hq, Hkv, Nq, Nkv, D = 2, 4, 2, 700, 1500, 512
  q, k, v = _rand((B, Hq, Nq, D), seed=501), _rand((B, Hkv, Nkv, D), seed=502), _rand((B, Hkv, Nkv, D), seed=503)
  rows, cols = torch.arange(Nq, device="cuda")[:, None], torch.arange(Nkv, device="cuda")[None, :]
  masks = {
      "causal": cols <= rows + (Nkv - Nq),
      "window": (cols <= rows + 400) & (cols >= rows + 100),
      "padding": (cols < 900).expand(Nq, Nkv),
      "holes": ((cols <= rows + 300) & (rows >= 64)),
  }
  for name, m in masks.items():
    m4 = m.view(1, 1, Nq, Nkv).contiguous()
    bb = hip.mask_kv_bounds(m4, Nq, Nkv)
    ba = hip.mask_kv_bounds(_additive(m4, q.dtype), Nq, Nkv)
    assert torch.equal(bb, ba), name
    o_all, l_all = hip.forward(q, k, v, m4, False, D ** -0.5, kv_bounds=False)
    o_clip, l_clip = hip.forward(q, k, v, m4, False, D ** -0.5, kv_bounds=True)
    assert _same_bits(o_all, o_clip) and _same_bits(l_all, l_clip), name
  # the unaligned / strided scan kernel
  odd = (cols <= rows + 300)[:, :1499].contiguous().view(1, 1, Nq, 1499)
  assert torch.equal(hip.mask_kv_bounds(odd, Nq, 1499), hip.mask_kv_bounds(_additive(odd, q.dtype), Nq, 1499))


@pytest.mark.parametrize("dtype", [torch.bool, torch.bfloat16, torch.float16, torch.float32])
def test_scan_kernel_ranges_equal_the_torch_reference(hip, dtype):
  """ffpa_attn_mask_kv_bounds (both the 16-byte and the element-wise kernels) against the same ranges computed with
  torch ops on the CPU: {first, end} of the keys any row of a 32-row block sees, {free_first, free_end} of the first run
  of keys on which the mask does nothing for all of them."""
  g = torch.Generator(device="cuda").manual_seed(5)
  for (bb, hb, nq, nkv) in ((1, 1, 700, 1536), (2, 3, 130, 1000), (1, 2, 64, 777), (1, 1, 1, 2048)):
    rows, cols = torch.arange(nq, device="cuda")[:, None], torch.arange(nkv, device="cuda")[None, :]
    keep = ((cols <= rows + nkv // 2) & (cols + 300 >= rows)).expand(bb, hb, nq, nkv).clone()
    keep &= torch.rand(bb, hb, 1, nkv, device="cuda", generator=g) > 0.02   # a few keys hidden everywhere: runs get cut
    if dtype == torch.bool:
      m = keep
    else:
      m = torch.zeros(bb, hb, nq, nkv, dtype=dtype, device="cuda").masked_fill(~keep, float("-inf"))
      m[:, :, :, nkv // 3] = 0.25  # a finite, non-neutral bias column: visible but not free
    want = hip.mask_kv_bounds(m.cpu(), nq, nkv)
    assert torch.equal(hip.mask_kv_bounds(m, nq, nkv).cpu(), want), (dtype, bb, hb, nq, nkv)
    wide = torch.zeros(bb, hb, nq, 2 * nkv, dtype=m.dtype, device="cuda")
    wide[..., ::2] = m
    assert torch.equal(hip.mask_kv_bounds(wide[..., ::2], nq, nkv).cpu(), want), ("strided", dtype)


def test_tiles_in_the_free_range_skip_the_mask_without_changing_a_bit(hip):
  """Results with the ranges (tile clipping + no mask reads in the neutral interior) == results walking and masking every
  tile, for boolean and additive masks, D <= 512 and split-D kernels; a forged free range over masked keys must show."""
  for D in (320, 512, 1024):
    B, Hq, Hkv, Nq, Nkv = 1, 4, 2, 900, 2048
    q, k, v = _rand((B, Hq, Nq, D), seed=701), _rand((B, Hkv, Nkv, D), seed=702), _rand((B, Hkv, Nkv, D), seed=703)
    rows, cols = torch.arange(Nq, device="cuda")[:, None], torch.arange(Nkv, device="cuda")[None, :]
    for name, keep in (("causal", cols <= rows + 600), ("window", (cols <= rows + 700) & (cols + 200 >= rows)), ("padding", (cols < 1500).expand(Nq, Nkv))):
      for m in (keep.view(1, 1, Nq, Nkv).contiguous(), _additive(keep.view(1, 1, Nq, Nkv), q.dtype)):
        o_all, l_all = hip.forward(q, k, v, m, False, D ** -0.5, kv_bounds=False)  # (both runs of a mask take the same build)
        bounds = hip.mask_kv_bounds(m, Nq, Nkv)
        assert int((bounds[..., 3] - bounds[..., 2]).max()) >= 512, name  # there IS an interior to skip
        o_rng, l_rng = hip.forward(q, k, v, m, False, D ** -0.5, kv_bounds=bounds)
        assert _same_bits(o_all, o_rng) and _same_bits(l_all, l_rng), (D, name, m.dtype)
    # the kernel really trusts the range: claim everything is free -> the mask is ignored -> equals the unmasked result
    forged = bounds.clone()
    forged[..., 0], forged[..., 1], forged[..., 2], forged[..., 3] = 0, Nkv, 0, Nkv
    o_plain, _ = hip.forward(q, k, v, None, False, D ** -0.5)
    for m in (m, keep.view(1, 1, Nq, Nkv).contiguous()):  # additive / boolean
      o_forged, _ = hip.forward(q, k, v, m, False, D ** -0.5, kv_bounds=forged)
      if D == 320 and m.dtype != torch.bool:  # (64-key tiles in the additive build, 128 in the unmasked one: equal up to rounding)
        _close(o_forged, o_plain, q.dtype, "forged")
      else:
        assert torch.equal(o_forged, o_plain), (D, m.dtype)


def test_public_api_bool_mask_allocates_nothing_mask_sized(hip):
  """ffpa_attn_func(attn_mask=<bool>) hands the mask's own bytes to the kernel: no 0 / -inf temporary (the reference
  allocates 2 B per mask element per call).  Peak memory during the call stays below the mask's own size."""
  from ffpa_attn_amd import ffpa_attn_func

  B, H, N, D = 1, 4, 8192, 512  # 256 row tiles: a full launch (no KV-split scratch); O + LSE = 32 MiB
  q, k, v = _rand((B, H, N, D), seed=71), _rand((B, H, N, D), seed=72), _rand((B, H, N, D), seed=73)
  mask = torch.ones(N, N, dtype=torch.bool, device="cuda").tril()  # 64 MiB; its additive bf16 form would be 128 MiB
  ffpa_attn_func(q, k, v, attn_mask=mask)  # warm-up (workspace caches, library load)
  torch.cuda.synchronize()
  torch.cuda.reset_peak_memory_stats()
  base = torch.cuda.memory_allocated()
  out = ffpa_attn_func(q, k, v, attn_mask=mask)
  torch.cuda.synchronize()
  extra = torch.cuda.max_memory_allocated() - base
  assert extra < mask.numel(), f"{extra} bytes allocated during the call, the mask itself is {mask.numel()}"
  ref = F.scaled_dot_product_attention(q, k, v, is_causal=True)
  _within_north_star(out, ref)


def test_baseline_config_4_through_the_public_api(hip):
  """BASELINE configs[3] as specified: B=2 Hq=32/Hkv=8 Nq=8192 Nkv=2048 D=320, GQA cross-attention with a causal mask —
  `ffpa_attn_func(attn_mask=tril, enable_gqa=True)` at full size (bool mask -> kernel bytes + automatic tile clipping),
  against SDPA's own top-left `is_causal` and against the op-level causal_offset=0 form."""
  from ffpa_attn_amd import ffpa_attn_func

  torch.manual_seed(0)
  q = torch.randn(2, 32, 8192, 320, dtype=torch.bfloat16, device="cuda")
  k = torch.randn(2, 8, 2048, 320, dtype=torch.bfloat16, device="cuda")
  v = torch.randn(2, 8, 2048, 320, dtype=torch.bfloat16, device="cuda")
  mask = torch.ones(8192, 2048, dtype=torch.bool, device="cuda").tril()
  out = ffpa_attn_func(q, k, v, attn_mask=mask, enable_gqa=True)
  ref = F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True)
  _within_north_star(out, ref)
  o0, _ = hip.forward(q, k, v, None, True, 320 ** -0.5, causal_offset=0)
  assert (out.float() - o0.float()).abs().max().item() <= 4e-3
  # the additive form of the same mask: the same values (its 64-key bias-tile build sums in another order than the 128-key tiles
  # of the boolean build, so to rounding, not to the bit)
  oa = ffpa_attn_func(q, k, v, attn_mask=_additive(mask, q.dtype), enable_gqa=True)
  assert (out.float() - oa.float()).abs().max().item() <= 8e-3


def test_backward_through_a_bool_mask(hip):
  from ffpa_attn_amd import ffpa_attn_func

  B, H, N, D = 1, 2, 640, 320
  mask = torch.rand(1, 1, N, N, device="cuda") > 0.2
  mask[..., 0] = True
  qs = [_rand((B, H, N, D), seed=s).requires_grad_(True) for s in (81, 82, 83)]
  rs = [t.detach().clone().requires_grad_(True) for t in qs]
  go = _rand((B, H, N, D), seed=84)
  ffpa_attn_func(*qs, attn_mask=mask).backward(go)
  s = (rs[0].float() @ rs[1].float().transpose(-1, -2)) * D ** -0.5
  (torch.softmax(s.masked_fill(~mask, float("-inf")), -1) @ rs[2].float()).to(go.dtype).backward(go)
  for a, b, n in zip(qs, rs, "qkv"):
    assert (a.grad.float() - b.grad.float()).abs().max().item() <= 3e-2 * max(1.0, b.grad.float().abs().max().item()), n


# ----------------------------------------------------------------------------- boundary hygiene (C-ABI v3)
def test_debug_kernels_live_only_in_the_test_library(hip):
  prod, dbg = hip.load_library(), hip.load_debug_library()
  assert prod.ffpa_attn_query(7) == 0 and dbg.ffpa_attn_query(7) == 1
  # FFPA_FLAG_DEBUG_SAFE_PATH against the product library: a clean status, not a launch
  q, k, v = _rand((1, 1, 64, 512)), _rand((1, 1, 64, 512), seed=1), _rand((1, 1, 64, 512), seed=2)
  o = torch.empty_like(q)
  p = hip.FfpaFwdParams()
  p.struct_size, p.abi_version = ctypes.sizeof(hip.FfpaFwdParams), hip.ABI_VERSION
  p.q, p.k, p.v, p.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
  p.batch, p.heads_q, p.heads_kv, p.seqlen_q, p.seqlen_kv, p.head_dim = 1, 1, 1, 64, 64, 512
  for name, t in (("q_stride", q), ("k_stride", k), ("v_stride", v), ("o_stride", o)):
    getattr(p, name)[:] = list(t.stride()[:3])
  p.softmax_scale, p.rescale_threshold, p.flags = 0.05, -1.0, hip.FLAG_DEBUG_SAFE_PATH
  rc = prod.ffpa_attn_fwd(ctypes.byref(p), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
  assert rc == 7, prod.ffpa_attn_last_error()  # FFPA_ERR_UNSUPPORTED
  assert dbg.ffpa_attn_fwd(ctypes.byref(p), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
  torch.cuda.synchronize()


def test_direct_op_calls_are_validated_before_pointers_reach_the_kernel(hip):
  q, k, v = _rand((1, 4, 64, 512)), _rand((1, 2, 640, 512), seed=1), _rand((1, 2, 640, 512), seed=2)
  with pytest.raises(ValueError, match="same shape"):
    hip.forward(q, k, v[:, :, :600], None, False, 0.05)
  with pytest.raises(ValueError, match="batch size and head dim"):
    hip.forward(q, k[..., :256], v[..., :256], None, False, 0.05)
  with pytest.raises(ValueError, match="multiple of key/value num_heads"):
    hip.forward(q[:, :3], k, v, None, False, 0.05)
  with pytest.raises(ValueError, match="batch size and head dim"):
    hip.forward(q, k.expand(2, -1, -1, -1), v.expand(2, -1, -1, -1), None, False, 0.05)
  with pytest.raises(ValueError, match="on q's device"):
    hip.forward(q, k, v, torch.zeros(1, 1, 64, 640), False, 0.05)
  # the op's real and fake implementations agree on layout for a padded head dim
  qq, kk, vv = _rand((1, 2, 520, 264)), _rand((1, 2, 520, 264), seed=1), _rand((1, 2, 520, 264), seed=2)
  o, _ = hip.forward(qq, kk, vv, None, False, 264 ** -0.5)
  assert o.is_contiguous() and o.shape == qq.shape


@pytest.mark.parametrize("D", [384, 512, 1024])
@pytest.mark.parametrize("bdtype", [torch.bfloat16, torch.float32])
def test_key_bias_row_cache_in_lds_is_bit_identical_to_the_global_reads(hip, D, bdtype):
  """A bias without a row axis ([B|1, H|1, 1, Nkv]: key padding as an additive mask, per-head key biases, the reference
  bench's 'attn-mask' case) is copied to LDS once per workgroup and read from there in every tile — same numbers as the
  per-tile global loads (FFPA_FLAG_NO_BIAS_LDS), also with a ragged last tile, -inf entries and GQA."""
  B, Hq, Hkv, Nq = 2, 4, 2, 300
  for Nkv, shape in ((1024, (1, 1, 1, 1024)), (1000, (2, 4, 1, 1000)), (2048, (1, 4, 1, 2048)), (1001, (1, 1, 1, 1001))):
    q, k, v = _rand((B, Hq, Nq, D), seed=D), _rand((B, Hkv, Nkv, D), seed=D + 1), _rand((B, Hkv, Nkv, D), seed=D + 2)
    g = torch.Generator(device="cuda").manual_seed(Nkv)
    bias = (torch.randn(shape, device="cuda", generator=g) * 0.5).to(bdtype)
    bias[..., 5::7] = float("-inf")
    o1, l1 = hip.forward(q, k, v, bias, False, D ** -0.5)
    o0, l0 = hip.forward(q, k, v, bias, False, D ** -0.5, flags=hip.FLAG_NO_BIAS_LDS)
    assert torch.equal(o1, o0) and torch.equal(l1, l0), (D, Nkv, shape)
    if bdtype == torch.bfloat16:
      _close(o1, F.scaled_dot_product_attention(q, k, v, attn_mask=bias, enable_gqa=True), q.dtype, f"sdpa {shape}")
    if D == 512 and Nkv == 1000:
      _check_vs_oracle(o1, l1, q, k, v, bias=_f32(bias.float()), name="key bias")


@pytest.mark.parametrize("D", [640, 768, 1024])
@pytest.mark.parametrize("bdtype", [torch.bfloat16, torch.float16, torch.float32])
def test_key_bias_ring_cache_of_the_pipelined_split_d_tiles(hip, D, bdtype):
  """Round 5: at D > 512 (D % 128 == 0) the key-bias build runs the softmax pipeline and reads its bias from a RING of fp32 entries (2048 at D = 1024: all the
  LDS has left) that is refilled half a ring at a time ahead of the walk.  Contexts of several ring lengths, ragged ends, -inf entries on both sides of every
  refill boundary, GQA, per-head biases: the same bits as the per-tile global loads (FFPA_FLAG_NO_BIAS_LDS: the additive-bias build), the oracle on a row sample."""
  B, Hq, Hkv, Nq = 1, 4, 2, 300
  dt = torch.float16 if bdtype == torch.float16 else torch.bfloat16
  for Nkv, shape in ((5000, (1, 1, 1, 5000)), (8192, (1, 4, 1, 8192)), (3073, (1, 1, 1, 3073)), (2049, (1, 4, 1, 2049))):
    q, k, v = _rand((B, Hq, Nq, D), dt, seed=D), _rand((B, Hkv, Nkv, D), dt, seed=D + 1), _rand((B, Hkv, Nkv, D), dt, seed=D + 2)
    g = torch.Generator(device="cuda").manual_seed(Nkv)
    bias = (torch.randn(shape, device="cuda", generator=g) * 0.5).to(bdtype)
    bias[..., 5::7] = float("-inf")
    for edge in (1023, 1024, 2047, 2048, 3071, 3072, 4095, 4096):
      if edge < Nkv:
        bias[..., edge] = float("-inf") if edge % 2 else 1.5
    plan = {}
    o1, l1 = hip.forward(q, k, v, bias, False, D ** -0.5, num_splits=1, plan_out=plan)
    assert f"ffpa_fwd_m16_kernel<{'fp16' if dt == torch.float16 else 'bf16'}, {D}, MK=3" in plan["kernel"], plan
    o0, l0 = hip.forward(q, k, v, bias, False, D ** -0.5, num_splits=1, flags=hip.FLAG_NO_BIAS_LDS, plan_out=plan)
    assert "MK=1" in plan["kernel"], plan
    assert torch.equal(o1, o0) and torch.equal(l1, l0), (D, Nkv, shape)
    if Nkv == 5000:
      _check_vs_oracle(o1[:, :2], l1[:, :2], q[:, :2], k[:, :1], v[:, :1], bias=_f32(bias.float()), rows=(100, 164), block_keys=32, name=f"ring D{D} {bdtype}")


@pytest.mark.parametrize("D", [64, 128, 256, 320, 384, 512, 640, 1024])
def test_bias_tiles_staged_through_lds_are_bit_identical_to_the_global_reads(hip, D):
  """A 16-bit bias WITH a row axis is LDS-DMA'd one KV step ahead into a private area per wave (the build with 64-key tiles at
  every head dim) and read there —
  same numbers as the per-tile global loads (FFPA_FLAG_NO_BIAS_LDS): broadcast batch / head dims, ragged rows and keys, -inf
  entries and fully hidden rows, causal on top, GQA, fp16."""
  B, Hq, Hkv = 2, 4, 2
  for (Nq, Nkv, shape, causal, dt) in ((300, 1024, (1, 1, 300, 1024), False, torch.bfloat16), (130, 1000, (2, 4, 130, 1000), False, torch.bfloat16),
                                      (513, 2048, (1, 4, 513, 2048), True, torch.bfloat16), (200, 520, (2, 1, 200, 520), False, torch.float16)):
    q, k, v = _rand((B, Hq, Nq, D), dt, seed=D), _rand((B, Hkv, Nkv, D), dt, seed=D + 1), _rand((B, Hkv, Nkv, D), dt, seed=D + 2)
    g = torch.Generator(device="cuda").manual_seed(Nkv + D)
    bias = (torch.randn(shape, device="cuda", generator=g) * 0.5).to(dt)
    bias[..., 3::11] = float("-inf")
    bias[:, :, 7, :] = float("-inf")  # a fully hidden row -> NaN
    o1, l1 = hip.forward(q, k, v, bias, causal, D ** -0.5, kv_bounds=False)
    o0, l0 = hip.forward(q, k, v, bias, causal, D ** -0.5, kv_bounds=False, flags=hip.FLAG_NO_BIAS_LDS)
    if D >= 320:  # (the 16x16x32 kernel: 64-key tiles whatever the bias source) the same fp32 values reach the same accumulators: the same bits
      assert _same_bits(o1, o0) and _same_bits(l1, l0), (D, Nq, Nkv, shape)
    else:         # D <= 256 (the 32x32x16 kernel): the global-read build uses 128-key tiles, another summation order
      keep = [r for r in range(Nq) if r != 7]
      _close(o1[:, :, keep], o0[:, :, keep], dt, f"tiles {D} {shape}")
      assert (l1[:, :, keep] - l0[:, :, keep]).abs().max().item() <= 2e-5
    assert torch.isnan(o1[:, :, 7]).all()
    if not causal:  # (PyTorch-ROCm's fused SDPA does not return NaN for the fully hidden row: compare the others)
      ref = F.scaled_dot_product_attention(q, k, v, attn_mask=bias, enable_gqa=True)
      keep = [r for r in range(Nq) if r != 7]
      _close(o1[:, :, keep], ref[:, :, keep], dt, f"sdpa {shape}")
    # with the mask ranges on top (tiles skipped / mask reads skipped): still the same bits
    o2, l2 = hip.forward(q, k, v, bias, causal, D ** -0.5, kv_bounds=True)
    assert _same_bits(o2, o1) and _same_bits(l2, l1)
  if D == 512:
    q, k, v = _rand((1, 2, 260, D), seed=1), _rand((1, 2, 700, D), seed=2), _rand((1, 2, 700, D), seed=3)
    bias = _rand((1, 2, 260, 704), seed=4)[..., :700]  # row stride 704 elements, 700 keys
    o1, l1 = hip.forward(q, k, v, bias, False, D ** -0.5)
    _check_vs_oracle(o1, l1, q, k, v, bias=_f32(bias.float()), name="bias tile, strided rows")


def test_bool_and_key_bias_masks_on_the_short_query_path(hip):
  """Nq <= 32 launches (split-KV tiles + LSE merge) with a boolean mask and with a key bias: the same bits as the additive /
  global-read forms, padding-style masks per batch element included."""
  D = 512
  for (B, Hq, Hkv, Nq, Nkv) in ((2, 8, 2, 1, 4096), (1, 4, 4, 7, 3000), (3, 4, 2, 20, 1500)):
    q, k, v = _rand((B, Hq, Nq, D), seed=Nq), _rand((B, Hkv, Nkv, D), seed=Nq + 1), _rand((B, Hkv, Nkv, D), seed=Nq + 2)
    lens = torch.tensor([Nkv - 17 * i for i in range(B)], device="cuda")
    keep = (torch.arange(Nkv, device="cuda")[None, :] < lens[:, None]).view(B, 1, 1, Nkv)  # key padding per batch element
    ob, lb = hip.forward(q, k, v, keep, False, D ** -0.5)
    oa, la = hip.forward(q, k, v, _additive(keep, q.dtype), False, D ** -0.5, flags=hip.FLAG_NO_BIAS_LDS)
    assert _same_bits(ob, oa) and _same_bits(lb, la), (Nq, Nkv)
    ref = F.scaled_dot_product_attention(q, k, v, attn_mask=keep, enable_gqa=True)
    _close(ob, ref, q.dtype, f"short-query bool {Nq}x{Nkv}")
    full = (torch.rand(B, Hq, Nq, Nkv, device="cuda") > 0.3)
    full[..., 0] = True
    ob2, _ = hip.forward(q, k, v, full, False, D ** -0.5)
    oa2, _ = hip.forward(q, k, v, _additive(full, q.dtype), False, D ** -0.5, flags=hip.FLAG_NO_BIAS_LDS)
    assert _same_bits(ob2, oa2)


def test_mask_created_under_inference_mode(hip, monkeypatch):
  """Masks built inside torch.inference_mode() (the usual serving path) have no version counter: reading `_version` raises.  The range
  scan must serve them with the cache on as well as off (round-3 advisor finding: the default-on cache crashed here)."""
  from ffpa_attn_amd import ffpa_attn_func

  B, H, N, D = 1, 4, 1024, 512
  q, k, v = _rand((B, H, N, D), seed=81), _rand((B, H, N, D), seed=82), _rand((B, H, N, D), seed=83)
  ref = F.scaled_dot_product_attention(q, k, v, is_causal=True)
  for cache in ("0", "1"):
    monkeypatch.setenv("FFPA_HIP_MASK_BOUNDS_CACHE", cache)
    with torch.inference_mode():
      mask = torch.ones(N, N, dtype=torch.bool, device="cuda").tril()
      out = ffpa_attn_func(q, k, v, attn_mask=mask)
      out2 = ffpa_attn_func(q, k, v, attn_mask=mask)
    assert torch.equal(out, out2)
    _within_north_star(out, ref)


def test_mask_range_cache_is_opt_in_and_follows_the_version_counter(hip, monkeypatch):
  """Default: every call scans the mask, so a write torch's version counter does not see (`mask.data`) is honoured.  Opt-in cache
  (FFPA_HIP_MASK_BOUNDS_CACHE=1): one scan per (tensor, version); an in-place torch write invalidates the entry; a consumer on
  another stream is ordered behind the scan's event."""
  B, H, N, D = 1, 2, 1024, 512
  q, k, v = _rand((B, H, N, D), seed=84), _rand((B, H, N, D), seed=85), _rand((B, H, N, D), seed=86)
  half = torch.zeros(1, 1, N, N, dtype=torch.bool, device="cuda")
  half[..., : N // 2] = True
  full = torch.ones(1, 1, N, N, dtype=torch.bool, device="cuda")
  o_half, _ = hip.forward(q, k, v, half, False, D ** -0.5, kv_bounds=False)
  o_full, _ = hip.forward(q, k, v, full, False, D ** -0.5, kv_bounds=False)

  monkeypatch.delenv("FFPA_HIP_MASK_BOUNDS_CACHE", raising=False)
  m = half.clone()
  assert torch.equal(hip.forward(q, k, v, m, False, D ** -0.5, kv_bounds=True)[0], o_half)
  m.data.fill_(True)  # invisible to the version counter
  assert torch.equal(hip.forward(q, k, v, m, False, D ** -0.5, kv_bounds=True)[0], o_full)

  monkeypatch.setenv("FFPA_HIP_MASK_BOUNDS_CACHE", "1")
  hip._BOUNDS_CACHE.clear()
  m = half.clone()
  assert torch.equal(hip.forward(q, k, v, m, False, D ** -0.5, kv_bounds=True)[0], o_half)
  assert len(hip._BOUNDS_CACHE) == 1
  first = next(iter(hip._BOUNDS_CACHE.values()))[2]
  assert hip.cached_mask_kv_bounds(m, N, N) is first  # served from the cache
  side = torch.cuda.Stream()
  with torch.cuda.stream(side):  # another stream: waits for the producer's event, same numbers
    o_side, _ = hip.forward(q, k, v, m, False, D ** -0.5, kv_bounds=True)
  side.synchronize()
  assert torch.equal(o_side, o_half)
  m.fill_(True)  # a torch in-place write: the counter moves, the entry is dropped
  assert torch.equal(hip.forward(q, k, v, m, False, D ** -0.5, kv_bounds=True)[0], o_full)
  hip._BOUNDS_CACHE.clear()
