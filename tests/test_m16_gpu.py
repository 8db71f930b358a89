"""The prefill kernel on the 16x16x32 MFMA shape (csrc/ffpa_fwd_m16_kernel.h; `pytest -m gpu`).

Unmasked and boolean-mask launches at head dims above 256 take this build by default; FFPA_FLAG_NO_M16 keeps them on the
32x32x16 build (same tiles, same recurrence, another summation order inside the matrix core).  Pinned here: the oracle (the
reference's recurrence restated on the CPU), the other build within output rounding, exact NaN / -inf patterns, and the
properties that must hold to the bit inside one build (determinism, head independence, KV splits merge, strided views).
"""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_fwd_gpu import _check_vs_oracle, _close, _f32, _rand, hip  # noqa: F401  (fixture + helpers)

pytestmark = pytest.mark.gpu

D = 512  # the headline head dim; DIMS: head dims this build is launched for (128-key tiles at 320, 64-key tiles to 512, split-D tiles above)
DIMS = [320, 384, 448, 512, 576, 640, 960, 1024]


def _both(hip, q, k, v, bias, causal, **kw):
  o16, l16 = hip.forward(q, k, v, bias, causal, q.size(-1) ** -0.5, **kw)
  o32, l32 = hip.forward(q, k, v, bias, causal, q.size(-1) ** -0.5, flags=hip.FLAG_NO_M16, **kw)
  return o16, l16, o32, l32


def _same_up_to_rounding(o16, l16, o32, l32, dtype, name):
  a, b = o16.float(), o32.float()
  assert torch.equal(torch.isnan(a), torch.isnan(b)), f"{name}: NaN pattern"
  fin = ~torch.isnan(a)
  # two correctly rounded-ish results of the same sums: at most a couple of storage ulps apart, on average far less
  ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
  d = (a - b).abs()[fin]
  scale = torch.maximum(b.abs(), torch.tensor(2.0 ** -6, device=b.device))[fin]
  assert (d <= 2 * ulp * scale + (2.5e-3 if dtype == torch.bfloat16 else 4e-4)).all(), f"{name}: max diff {d.max().item():.3e}"
  assert d.mean().item() <= 0.5 * ulp * scale.mean().item() + 1e-5, f"{name}: mean diff {d.mean().item():.3e}"
  if l16 is not None:
    assert torch.equal(torch.isinf(l16), torch.isinf(l32)) and torch.equal(torch.isnan(l16), torch.isnan(l32)), f"{name}: LSE pattern"
    ok = torch.isfinite(l32)
    assert (l16[ok] - l32[ok]).abs().max().item() <= 2e-5, name


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D", DIMS)
@pytest.mark.parametrize("case", [(1, 2, 2, 128, 64, False), (1, 2, 1, 200, 333, False), (2, 4, 2, 384, 384, True), (1, 2, 2, 77, 1000, True),
                                  (1, 1, 1, 640, 1500, False), (1, 4, 4, 33, 65, True), (1, 2, 2, 1, 64, False), (2, 2, 1, 129, 63, False)])
def test_matches_the_oracle_and_the_other_build(hip, dtype, D, case):
  B, Hq, Hkv, Nq, Nkv, causal = case
  q, k, v = _rand((B, Hq, Nq, D), dtype, seed=Nq), _rand((B, Hkv, Nkv, D), dtype, seed=Nkv + 1), _rand((B, Hkv, Nkv, D), dtype, seed=Nkv + 2)
  o16, l16, o32, l32 = _both(hip, q, k, v, None, causal, num_splits=1)
  _same_up_to_rounding(o16, l16, o32, l32, dtype, str(case))
  _check_vs_oracle(o16, l16, q, k, v, causal=causal, block_keys=hip.tile_config(D)["block_keys"], name=f"m16 D{D} {case}")


def test_the_default_launch_is_this_build(hip):
  """The two builds sum in different orders: on random data their outputs cannot agree in every bit — if they do, the flag (or
  the dispatch) is not doing anything."""
  q, k, v = _rand((1, 4, 512, D), seed=1), _rand((1, 4, 2048, D), seed=2), _rand((1, 4, 2048, D), seed=3)
  o16, _, o32, _ = _both(hip, q, k, v, None, False)
  assert not torch.equal(o16, o32)
  mask = torch.ones(1, 1, 512, 2048, dtype=torch.bool, device="cuda")
  ob16, _, ob32, _ = _both(hip, q, k, v, mask, False, kv_bounds=False)
  assert torch.equal(ob16, o16) and torch.equal(ob32, o32)  # an all-True mask changes nothing, in either build
  # additive biases stay on the 32x32x16 build whatever the flag says
  zero = torch.zeros(1, 1, 512, 2048, dtype=q.dtype, device="cuda")
  oz, _ = hip.forward(q, k, v, zero, False, D ** -0.5, kv_bounds=False)
  assert torch.equal(oz, o32)


@pytest.mark.parametrize("D", [320, 512, 1024])
def test_causal_offsets_tails_and_fully_masked_rows(hip, D):
  Nq, Nkv = 300, 700
  q, k, v = _rand((1, 2, Nq, D), seed=11), _rand((1, 2, Nkv, D), seed=12), _rand((1, 2, Nkv, D), seed=13)
  for off in (0, 400, -40, 650):  # SDPA-style, tail-aligned, rows with no visible key (NaN), almost everything visible
    o16, l16 = hip.forward(q, k, v, None, True, D ** -0.5, causal_offset=off)
    o32, l32 = hip.forward(q, k, v, None, True, D ** -0.5, causal_offset=off, flags=hip.FLAG_NO_M16)
    _same_up_to_rounding(o16, l16, o32, l32, q.dtype, f"offset {off}")
    _check_vs_oracle(o16, l16, q, k, v, causal=True, causal_offset=off, block_keys=hip.tile_config(D)["block_keys"], name=f"m16 offset {off}")
  assert torch.isnan(o16).sum().item() == 0
  o, lse = hip.forward(q, k, v, None, True, D ** -0.5, causal_offset=-40)
  assert torch.isnan(o[:, :, :40]).all() and not torch.isnan(o[:, :, 40:]).any()
  assert torch.equal(o[0, :, 40], v[0, :, 0])  # a row that sees exactly one key returns that key's V row


@pytest.mark.parametrize("D", DIMS)
def test_boolean_masks_vector_and_byte_paths(hip, D):
  """Mask bytes are read 4 at a time (unit key stride, 16-byte aligned rows, full tile) or one by one; ranges skip tiles and
  mask reads.  Same visible keys as the 32x32x16 build: same NaN rows, outputs equal up to rounding; with and without the
  ranges the SAME bits."""
  for Nq, Nkv in ((513, 1024), (513, 1000), (130, 777)):
    q, k, v = _rand((1, 2, Nq, D), seed=61), _rand((1, 2, Nkv, D), seed=62), _rand((1, 2, Nkv, D), seed=63)
    g = torch.Generator(device="cuda").manual_seed(Nq + Nkv)
    mask = torch.rand(1, 2, Nq, Nkv, device="cuda", generator=g) > 0.3
    mask[0, 0, 5, :] = False
    mask[0, :, 9, :128] = False
    mask[0, 1, 100, 1:] = False
    mask[0, 1, 100, 0] = True
    o16, l16, o32, l32 = _both(hip, q, k, v, mask, False, kv_bounds=False)
    _same_up_to_rounding(o16, l16, o32, l32, q.dtype, f"bool {Nq}x{Nkv}")
    assert torch.isnan(o16[0, 0, 5]).all() and torch.equal(o16[0, 1, 100], v[0, 1, 0])
    for m in (mask, mask[:, :1], mask[:, :, :1], mask[..., ::2].repeat_interleave(2, -1)[..., :Nkv]):  # broadcast heads / rows, odd strides
      a, la = hip.forward(q, k, v, m, False, D ** -0.5, kv_bounds=False)
      b, lb = hip.forward(q, k, v, m, False, D ** -0.5, kv_bounds=True)
      assert torch.equal(torch.nan_to_num(a.float(), nan=7.0), torch.nan_to_num(b.float(), nan=7.0)), (Nq, Nkv, tuple(m.shape))
      _close(a[~torch.isnan(a).any(-1)], F.scaled_dot_product_attention(q, k, v, attn_mask=m)[~torch.isnan(a).any(-1)], q.dtype, "sdpa")
  # structured masks: the interior of the visible band is never read, tiles outside are never visited
  Nq, Nkv = 900, 2048
  q, k, v = _rand((1, 4, Nq, D), seed=701), _rand((1, 2, Nkv, D), seed=702), _rand((1, 2, Nkv, D), seed=703)
  rows, cols = torch.arange(Nq, device="cuda")[:, None], torch.arange(Nkv, device="cuda")[None, :]
  for name, keep in (("causal", cols <= rows + 600), ("window", (cols <= rows + 700) & (cols + 200 >= rows)), ("padding", (cols < 1500).expand(Nq, Nkv))):
    m = keep.view(1, 1, Nq, Nkv).contiguous()
    a, la = hip.forward(q, k, v, m, False, D ** -0.5, kv_bounds=False)
    b, lb = hip.forward(q, k, v, m, False, D ** -0.5, kv_bounds=True)
    assert torch.equal(a, b) and torch.equal(la, lb), name
    if name == "causal":
      c, lc = hip.forward(q, k, v, None, True, D ** -0.5, causal_offset=600)
      assert torch.equal(a, c) and torch.equal(la, lc)  # the explicit mask and the structural one: the same numbers


def test_determinism_head_independence_and_strided_views(hip):
  B, Hq, Hkv, N = 2, 4, 2, 520
  q, k, v = _rand((B, Hq, N, D), seed=21), _rand((B, Hkv, N, D), seed=22), _rand((B, Hkv, N, D), seed=23)
  o, lse = hip.forward(q, k, v, None, True, D ** -0.5)
  o2, lse2 = hip.forward(q, k, v, None, True, D ** -0.5)
  assert torch.equal(o, o2) and torch.equal(lse, lse2)
  for b, h in ((0, 0), (1, 3)):
    o1, l1 = hip.forward(q[b:b + 1, h:h + 1], k[b:b + 1, h // 2:h // 2 + 1], v[b:b + 1, h // 2:h // 2 + 1], None, True, D ** -0.5)
    assert torch.equal(o[b:b + 1, h:h + 1], o1) and torch.equal(lse[b:b + 1, h:h + 1], l1), (b, h)
  # token-major storage ([B, N, H, D] viewed as [B, H, N, D]) and a K/V cache slice: strides, not copies
  qt = q.transpose(1, 2).contiguous().transpose(1, 2)
  kt = k.transpose(1, 2).contiguous().transpose(1, 2)
  vt = v.transpose(1, 2).contiguous().transpose(1, 2)
  ot, lt = hip.forward(qt, kt, vt, None, True, D ** -0.5)
  assert torch.equal(ot, o) and torch.equal(lt, lse)
  kc, vc = torch.cat([k, k], 2)[:, :, :N], torch.cat([v, v], 2)[:, :, :N]
  oc, _ = hip.forward(q, kc, vc, None, True, D ** -0.5)
  assert torch.equal(oc, o)
  # scaling V scales O exactly (powers of two)
  o4, _ = hip.forward(q, k, v * 4, None, True, D ** -0.5)
  assert torch.equal(o4, o * 4)


def test_underfilled_launch_splits_the_kv_axis_and_merges(hip):
  """Few row tiles against a long context: the plan splits the KV tiles over workgroups (fp32 partials + LSE, merged by the
  merge kernel) — the split result equals the unsplit one up to the fp32 merge."""
  q, k, v = _rand((1, 4, 256, D), seed=31), _rand((1, 4, 16384, D), seed=32), _rand((1, 4, 16384, D), seed=33)
  plan = {}
  o_s, l_s = hip.forward(q, k, v, None, False, D ** -0.5, plan_out=plan)
  assert plan["splits"] > 1
  o_1, l_1 = hip.forward(q, k, v, None, False, D ** -0.5, num_splits=1)
  assert (o_s.float() - o_1.float()).abs().max().item() <= 2.0 ** -8 and (l_s - l_1).abs().max().item() <= 1e-5
  _check_vs_oracle(o_s, l_s, q, k, v, rows=(0, 64), block_keys=64, name="m16 split")


@pytest.mark.parametrize("d", [264, 328, 456, 504, 520, 968, 1016])
def test_ragged_head_dims_equal_the_padded_run(hip, d):
  """Head dims between the built multiples of 64: missing columns read as zeros in-kernel — the same bits as the host-padded run
  of this build."""
  D = hip.padded_head_dim(d)
  for (B, Hq, Hkv, Nq, Nkv, causal) in ((1, 2, 1, 130, 257, False), (1, 2, 2, 200, 333, True)):
    q, k, v = _rand((B, Hq, Nq, d), seed=d), _rand((B, Hkv, Nkv, d), seed=d + 1), _rand((B, Hkv, Nkv, d), seed=d + 2)
    o, lse = hip.forward(q, k, v, None, causal, d ** -0.5)
    qp, kp, vp = (F.pad(t, (0, D - d)) for t in (q, k, v))
    op, lsep = hip.forward(qp, kp, vp, None, causal, d ** -0.5)
    assert torch.equal(o, op[..., :d]) and torch.equal(lse, lsep), (d, Nq, Nkv, causal)
    o32, l32 = hip.forward(q, k, v, None, causal, d ** -0.5, flags=hip.FLAG_NO_M16)
    _same_up_to_rounding(o, lse, o32, l32, q.dtype, f"d{d}")


def test_public_api_headline_shape_slice(hip):
  """ffpa_attn_func at the headline head dim goes through this build and agrees with SDPA within the north star's tolerance."""
  from ffpa_attn_amd import ffpa_attn_func

  q, k, v = _rand((1, 8, 2048, D), seed=41), _rand((1, 8, 2048, D), seed=42), _rand((1, 8, 2048, D), seed=43)
  out = ffpa_attn_func(q, k, v)
  direct, _ = hip.forward(q, k, v, None, False, D ** -0.5)
  assert torch.equal(out, direct)
  ref = F.scaled_dot_product_attention(q, k, v)
  assert (out.float() - ref.float()).abs().max().item() <= 1e-2
  outc = ffpa_attn_func(q, k, v, is_causal=True)
  assert (outc.float() - F.scaled_dot_product_attention(q, k, v, is_causal=True).float()).abs().max().item() <= 1e-2


@pytest.mark.parametrize("D", [320, 512, 1024])
def test_dropout_keeps_the_same_scores_as_the_other_build(hip, D):
  """Dropout launches take this build too (Philox at the logical score index: the kept set does not depend on the lane layout), with
  and without a mask or an additive bias riding along — the only launches of its additive-bias code.  Same kept scores as the
  32x32x16 build: outputs equal up to rounding (one flipped keep decision would move O by ~ p_ij / (1 - p) ~ 1e-1 at these sizes)."""
  B, Hq, Hkv, Nq, Nkv = 1, 4, 2, 200, 333
  q, k, v = _rand((B, Hq, Nq, D), seed=81), _rand((B, Hkv, Nkv, D), seed=82), _rand((B, Hkv, Nkv, D), seed=83)
  g = torch.Generator(device="cuda").manual_seed(5)
  mask = torch.rand(1, 1, Nq, Nkv, device="cuda", generator=g) > 0.2
  mask[..., 0] = True
  cases = {"plain": None, "bool": mask, "f32": torch.randn(1, Hq, Nq, Nkv, device="cuda", generator=g) * 0.5,
           "key_bias": (torch.randn(1, 1, 1, Nkv, device="cuda", generator=g) * 0.5).to(q.dtype),
           "bf16_strided": (torch.randn(1, 1, Nq, 2 * Nkv, device="cuda", generator=g) * 0.5).to(q.dtype)[..., ::2]}
  for name, bias in cases.items():
    for causal in (False, True):
      kw = dict(dropout_p=0.3, philox_seed=99, philox_offset=12, kv_bounds=False)
      o16, l16 = hip.forward(q, k, v, bias, causal, D ** -0.5, **kw)
      o32, l32 = hip.forward(q, k, v, bias, causal, D ** -0.5, flags=hip.FLAG_NO_M16, **kw)
      _same_up_to_rounding(o16, l16, o32, l32, q.dtype, f"dropout {name} causal={causal}")
      o_nodrop, _ = hip.forward(q, k, v, bias, causal, D ** -0.5, kv_bounds=False)
      assert (o16.float() - o_nodrop.float()).abs().max().item() > 0.02


@pytest.mark.parametrize("D", [128, 512, 1024])
def test_caller_packed_query_heads_at_prefill_sizes(hip, D):
  """ffpa_fwd_params.causal_row_mod with more than 32 packed rows (prefill tiles, both builds): a caller that packs the query heads
  of a KV group into the row axis gets, row for row, the bits of the unpacked call — each row's recurrence is the same, only its
  place in a tile differs."""
  B, Hkv, g, Nq, Nkv = 2, 2, 4, 40, 333
  q = _rand((B, Hkv * g, Nq, D), seed=91)
  k, v = _rand((B, Hkv, Nkv, D), seed=92), _rand((B, Hkv, Nkv, D), seed=93)
  for off in (Nkv - Nq, 0, 100):
    o_ref, l_ref = hip.forward(q, k, v, None, True, D ** -0.5, causal_offset=off, num_splits=1)
    qp = q.view(B, Hkv, g * Nq, D)
    o, lse = hip.forward(qp, k, v, None, True, D ** -0.5, causal_offset=off, causal_row_mod=Nq, num_splits=1)
    assert torch.equal(o.view(B, Hkv * g, Nq, D), o_ref) and torch.equal(lse.view(B, Hkv * g, Nq), l_ref), (D, off)


def test_non_positive_scale_takes_the_other_build(hip):
  """This build folds softmax_scale into the exponent's FMA and scales the row max after its reduction — exact only for a positive
  scale; zero / negative scales (legal, if unusual) are served by the 32x32x16 build: same bits as with FFPA_FLAG_NO_M16, and right."""
  q, k, v = _rand((1, 2, 200, D), seed=101), _rand((1, 2, 333, D), seed=102), _rand((1, 2, 333, D), seed=103)
  for scale in (-0.044, 0.0):
    o, lse = hip.forward(q, k, v, None, True, scale)
    o32, l32 = hip.forward(q, k, v, None, True, scale, flags=hip.FLAG_NO_M16)
    assert torch.equal(o, o32) and torch.equal(lse, l32), scale
    s = (q.float() @ k.float().transpose(-1, -2)) * scale
    r, c = torch.arange(200, device="cuda")[:, None], torch.arange(333, device="cuda")[None, :]
    s = s.masked_fill(c > r + 133, float("-inf"))
    want = torch.softmax(s, -1) @ v.float()
    assert (o.float() - want).abs().max().item() < 1e-2, scale
    assert (lse - torch.logsumexp(s, -1)).abs().max().item() < 1e-3, scale
