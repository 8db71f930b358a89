"""The prefill kernel on the 16x16x32 MFMA shape (csrc/ffpa_fwd_m16_kernel.h; `pytest -m gpu`).

EVERY prefill launch at head dims >= 128 runs this kernel (unmasked, boolean masks, additive biases, dropout, any softmax scale).
Pinned here: the oracle (the reference's recurrence restated on the CPU), the register-staged 32x32x16 twin of the test library
(another mapping of the same tiles onto the matrix core: same recurrence, another summation order) within output rounding where
that twin exists (bf16; head dims 128 / 320 / 512 / 640 / 1024), exact NaN / -inf patterns, and the properties that must hold to the bit
inside one build (determinism, head independence, KV splits merge, strided views, every bias source giving the same bits).
"""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_fwd_gpu import _check_vs_oracle, _close, _f32, _rand, hip  # noqa: F401  (fixture + helpers)

pytestmark = pytest.mark.gpu

D = 512  # the headline head dim; DIMS: head dims this build is launched for (128-key tiles at 320, 64-key tiles to 512, split-D tiles above)
DIMS = [128, 192, 256, 320, 384, 448, 512, 576, 640, 960, 1024]


TWIN_DIMS = (128, 320, 512, 640, 1024)  # head dims whose register-staged 32x32x16 twin is built into libffpa_attn_hip_test.so (bf16 only)


def _has_twin(q):
  return q.dtype == torch.bfloat16 and q.size(-1) in TWIN_DIMS


def _twin(hip, q, k, v, bias, causal, scale=None, **kw):
  """The same launch on the register-staged 32x32x16 twin kernel of the test library."""
  kw.pop("num_splits", None)
  return hip.forward(q, k, v, bias, causal, q.size(-1) ** -0.5 if scale is None else scale, flags=hip.FLAG_DEBUG_SAFE_PATH, num_splits=1, **kw)


def _both(hip, q, k, v, bias, causal, **kw):
  o16, l16 = hip.forward(q, k, v, bias, causal, q.size(-1) ** -0.5, **kw)
  o32, l32 = _twin(hip, q, k, v, bias, causal, **kw) if _has_twin(q) else (None, None)
  return o16, l16, o32, l32


def _same_up_to_rounding(o16, l16, o32, l32, dtype, name):
  if o32 is None:  # no twin for this dtype / head dim
    return
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
def test_matches_the_oracle_and_the_twin_build(hip, dtype, D, case):
  B, Hq, Hkv, Nq, Nkv, causal = case
  q, k, v = _rand((B, Hq, Nq, D), dtype, seed=Nq), _rand((B, Hkv, Nkv, D), dtype, seed=Nkv + 1), _rand((B, Hkv, Nkv, D), dtype, seed=Nkv + 2)
  o16, l16, o32, l32 = _both(hip, q, k, v, None, causal, num_splits=1)
  _same_up_to_rounding(o16, l16, o32, l32, dtype, str(case))
  _check_vs_oracle(o16, l16, q, k, v, causal=causal, block_keys=hip.tile_config(D)["block_keys"], name=f"m16 D{D} {case}")


def test_the_launch_plan_names_this_build(hip):
  """The C-ABI says which kernel a launch runs (ffpa_attn_fwd_kernel): every prefill launch at D >= 128 names ffpa_fwd_m16_kernel with
  the mask kind of its build; the two mappings sum in different orders, so on random data the twin cannot agree in every bit."""
  q, k, v = _rand((1, 4, 512, D), seed=1), _rand((1, 4, 2048, D), seed=2), _rand((1, 4, 2048, D), seed=3)
  plan = {}
  o16, _ = hip.forward(q, k, v, None, False, D ** -0.5, plan_out=plan)
  assert plan["splits"] > 1 and plan["kernel"] == "ffpa_fwd_m16_kernel<bf16, 512, MK=0, DROP=0> + ffpa_fwd_merge_kernel", plan  # (4 row tiles: the KV axis is split)
  o16, _ = hip.forward(q, k, v, None, False, D ** -0.5, plan_out=plan, num_splits=1)
  assert plan["kernel"] == "ffpa_fwd_m16_kernel<bf16, 512, MK=0, DROP=0>", plan
  o32, _ = _twin(hip, q, k, v, None, False)
  assert not torch.equal(o16, o32)
  mask = torch.ones(1, 1, 512, 2048, dtype=torch.bool, device="cuda")
  ob16, _ = hip.forward(q, k, v, mask, False, D ** -0.5, kv_bounds=False, plan_out=plan, num_splits=1)
  assert plan["kernel"] == "ffpa_fwd_m16_kernel<bf16, 512, MK=2, DROP=0>", plan
  assert torch.equal(ob16, o16)  # an all-True mask changes nothing
  zero = torch.zeros(1, 1, 512, 2048, dtype=q.dtype, device="cuda")
  oz, _ = hip.forward(q, k, v, zero, False, D ** -0.5, kv_bounds=False, plan_out=plan, num_splits=1)
  assert plan["kernel"] == "ffpa_fwd_m16_kernel<bf16, 512, MK=1, DROP=0>" and plan["block_keys"] == 64, plan
  assert torch.equal(oz, o16)  # an all-zero bias: the accumulators start from 0.0 either way
  hip.forward(q, k, v, None, False, D ** -0.5, dropout_p=0.1, philox_seed=1, plan_out=plan, num_splits=1)
  assert plan["kernel"] == "ffpa_fwd_m16_kernel<bf16, 512, MK=0, DROP=1>", plan
  hip.forward(q, k, v, mask, False, D ** -0.5, dropout_p=0.1, philox_seed=1, kv_bounds=False, plan_out=plan, num_splits=1)
  assert plan["kernel"] == "ffpa_fwd_m16_kernel<bf16, 512, MK=1, DROP=1>", plan
  hip.forward(q[..., :256].contiguous(), k[..., :256].contiguous(), v[..., :256].contiguous(), None, False, 256 ** -0.5, plan_out=plan, num_splits=1)
  assert plan["kernel"] == "ffpa_fwd_m16_kernel<bf16, 256, MK=0, DROP=0>", plan
  hip.forward(q[..., :64].contiguous(), k[..., :64].contiguous(), v[..., :64].contiguous(), None, False, 64 ** -0.5, plan_out=plan, num_splits=1)
  assert plan["kernel"].startswith("ffpa_fwd_split_d_kernel<bf16, 64, ND=1"), plan  # (D = 64: the 32x32x16 kernel)
  hip.forward(q[:, :, :1], k, v, None, False, D ** -0.5, plan_out=plan)
  assert plan["kernel"].startswith("ffpa_fwd_split_d_kernel<bf16, 512, ND=4") and plan["kernel"].endswith("+ ffpa_fwd_merge_kernel"), plan
  hip.forward(q[:, :, :1], k, v, None, False, D ** -0.5, plan_out=plan, merge_in_launch=True)
  assert plan["kernel"].endswith("(in-launch split merge)"), plan
  kb = _rand((1, 1, 1, 2048), seed=9)
  hip.forward(q, k, v, kb, False, D ** -0.5, plan_out=plan, num_splits=1)
  assert plan["kernel"] == "ffpa_fwd_m16_kernel<bf16, 512, MK=3, DROP=0>" and plan["block_keys"] == 64, plan  # a key bias: the lean build


@pytest.mark.parametrize("D", [320, 512, 1024])
def test_causal_offsets_tails_and_fully_masked_rows(hip, D):
  Nq, Nkv = 300, 700
  q, k, v = _rand((1, 2, Nq, D), seed=11), _rand((1, 2, Nkv, D), seed=12), _rand((1, 2, Nkv, D), seed=13)
  for off in (0, 400, -40, 650):  # SDPA-style, tail-aligned, rows with no visible key (NaN), almost everything visible
    o16, l16 = hip.forward(q, k, v, None, True, D ** -0.5, causal_offset=off)
    o32, l32 = _twin(hip, q, k, v, None, True, causal_offset=off)
    _same_up_to_rounding(o16, l16, o32, l32, q.dtype, f"offset {off}")
    _check_vs_oracle(o16, l16, q, k, v, causal=True, causal_offset=off, block_keys=hip.tile_config(D)["block_keys"], name=f"m16 offset {off}")
  assert torch.isnan(o16).sum().item() == 0
  o, lse = hip.forward(q, k, v, None, True, D ** -0.5, causal_offset=-40)
  assert torch.isnan(o[:, :, :40]).all() and not torch.isnan(o[:, :, 40:]).any()
  assert torch.equal(o[0, :, 40], v[0, :, 0])  # a row that sees exactly one key returns that key's V row


@pytest.mark.parametrize("D", DIMS)
def test_boolean_masks_vector_and_byte_paths(hip, D):
  """Mask bytes are read 4 at a time (unit key stride, 16-byte aligned rows, full tile) or one by one; ranges skip tiles and
  mask reads.  Same visible keys as the 32x32x16 twin: same NaN rows, outputs equal up to rounding; with and without the
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
      # the explicit mask and the structural one: the same numbers from the same launch plan (this small launch under-fills the chip: left to itself the
      # plan splits the KV axis, and since round 5 it knows from the causal FLAG that the keys past row 899 + 600 are hidden from every row — no KV range
      # out there —, which it cannot know of a mask: num_splits = 1 on both sides)
      a1, la1 = hip.forward(q, k, v, m, False, D ** -0.5, kv_bounds=False, num_splits=1)
      c, lc = hip.forward(q, k, v, None, True, D ** -0.5, causal_offset=600, num_splits=1)
      assert torch.equal(a1, c) and torch.equal(la1, lc)


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
  _check_vs_oracle(o_s, l_s, q, k, v, rows=(0, 64), block_keys=64, name="m16 split", split=True)


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
    # O^T columns past the caller's head dim are exact zeros in the accumulators (K AND V slots there are zero-filled by the range
    # check): the padded run must not see anything but zeros in the columns it drops
    assert not torch.isnan(op[..., d:].float()).any() and (op[..., d:] == 0).all(), (d, Nq, Nkv, causal)


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


def _dropout_reference(q, k, v, bias, causal, p, seed, offset, scale):
  """fp32 math with the kernel's dropout convention (ffpa_attn_amd/philox.py: the torch-ops Philox the backward uses to rebuild the
  forward's mask): softmax first, then the keep mask on the rounded P, 1 / (1 - p) scaling."""
  from ffpa_attn_amd.philox import dropout_keep_mask

  B, Hq, Nq, _ = q.shape
  Hkv, Nkv = k.size(1), k.size(2)
  g = Hq // Hkv
  s = (q.float() @ k.float().repeat_interleave(g, 1).transpose(-1, -2)) * scale
  if bias is not None:
    s = s.masked_fill(~bias, float("-inf")) if bias.dtype == torch.bool else s + bias.float()
  if causal:
    r, c = torch.arange(Nq, device=q.device)[:, None], torch.arange(Nkv, device=q.device)[None, :]
    s = s.masked_fill(c > r + (Nkv - Nq), float("-inf"))
  pr = torch.softmax(s, -1).to(q.dtype).float()
  idx = torch.arange(B * Hq * Nq * Nkv, device=q.device, dtype=torch.int64).view(B, Hq, Nq, Nkv)
  keep = dropout_keep_mask(seed, offset, idx, p)
  return ((pr * keep / (1.0 - p)).to(q.dtype).float() @ v.float().repeat_interleave(g, 1)), torch.logsumexp(s, -1)


@pytest.mark.parametrize("D", [320, 512, 1024])
def test_dropout_builds_with_and_without_a_bias(hip, D):
  """Dropout launches: the bias-free build (MK = 0) when nothing rides along, the additive-bias build with a mask or a bias.  Philox at
  the logical score index: the kept set does not depend on the lane layout — checked against fp32 math with the same keep mask (one
  flipped keep decision would move O by ~ p_ij / (1 - p) ~ 1e-1 at these sizes)."""
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
      for off in (12, 13):  # Philox groups aligned / straddling two blocks
        kw = dict(dropout_p=0.3, philox_seed=99, philox_offset=off, kv_bounds=False)
        o16, l16 = hip.forward(q, k, v, bias, causal, D ** -0.5, **kw)
        want, lse_want = _dropout_reference(q, k, v, bias, causal, 0.3, 99, off, D ** -0.5)
        d = (o16.float() - want).abs()
        assert d.max().item() <= 2.5e-2 and d.mean().item() <= 1.5e-3, (name, causal, off, d.max().item(), d.mean().item())
        assert (l16 - lse_want).abs().max().item() <= 2e-4, (name, causal)
        o_nodrop, _ = hip.forward(q, k, v, bias, causal, D ** -0.5, kv_bounds=False)
        assert (o16.float() - o_nodrop.float()).abs().max().item() > 0.02
  # a non-positive softmax scale next to dropout (the multiply-first form / the zeroed-Q form of the kernel)
  for scale in (-0.05, 0.0):
    o16, _ = hip.forward(q, k, v, cases["key_bias"], False, scale, dropout_p=0.3, philox_seed=99, philox_offset=12, kv_bounds=False)
    want, _ = _dropout_reference(q, k, v, cases["key_bias"], False, 0.3, 99, 12, scale)
    assert (o16.float() - want).abs().max().item() <= 2.5e-2, scale


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


def test_non_positive_softmax_scales(hip):
  """The kernel folds softmax_scale into the exponent's FMA and scales the row max after its reduction — exact only for a positive
  scale.  A negative scale takes its multiply-first form (a wave-uniform branch), a zero scale reaches it as "Q = 0, scale = 1" (the
  same scores: 0 * q.k + bias): legal, if unusual — and right, with and without a bias."""
  q, k, v = _rand((1, 2, 200, D), seed=101), _rand((1, 2, 333, D), seed=102), _rand((1, 2, 333, D), seed=103)
  g = torch.Generator(device="cuda").manual_seed(7)
  biases = (None, (torch.randn(1, 2, 200, 333, device="cuda", generator=g)).to(q.dtype), torch.randn(1, 1, 1, 333, device="cuda", generator=g))
  for scale in (-0.044, 0.0):
    for bias in biases:
      o, lse = hip.forward(q, k, v, bias, True, scale, kv_bounds=False)
      s = (q.float() @ k.float().transpose(-1, -2)) * scale
      if bias is not None:
        s = s + bias.float()
      r, c = torch.arange(200, device="cuda")[:, None], torch.arange(333, device="cuda")[None, :]
      s = s.masked_fill(c > r + 133, float("-inf"))
      want = torch.softmax(s, -1) @ v.float()
      assert (o.float() - want).abs().max().item() < 1e-2, (scale, None if bias is None else tuple(bias.shape))
      assert (lse - torch.logsumexp(s, -1)).abs().max().item() < 1e-3, scale
      if bias is None and scale < 0:
        o32, l32 = _twin(hip, q, k, v, None, True, scale=scale)
        _same_up_to_rounding(o, lse, o32, l32, q.dtype, f"scale {scale}")


def _bias_cases(B, Hq, Nq, Nkv, dtype, gen):
  """Additive biases by where the kernel takes them from: key biases (LDS row cache), row-axis biases with unit key stride and 16-byte
  aligned rows (LDS-DMA staged tiles), everything else (element loads)."""
  def r(*shape, dt=dtype):
    return (torch.randn(*shape, device="cuda", generator=gen) * 0.7).to(dt)

  wide = r(1, 1, Nq, 2 * Nkv + 8)
  cases = {
    "key_bias": r(1, 1, 1, Nkv),
    "key_bias_per_head_f32": r(B, Hq, 1, Nkv, dt=torch.float32),
    "key_bias_strided": r(1, 1, 1, 2 * Nkv)[..., ::2],
    "dense": r(1, 1, Nq, Nkv),
    "dense_per_head": r(B, Hq, Nq, Nkv),
    "dense_f32": r(1, Hq, Nq, Nkv, dt=torch.float32),
    "dense_key_strided": wide[..., : 2 * Nkv : 2],
    "dense_row_padded": wide[..., 8 : 8 + Nkv] if Nkv % 8 == 0 else wide[..., 1 : 1 + Nkv],
    "dense_misaligned": wide[..., 1 : 1 + Nkv],
    "row_bias": r(1, 1, Nq, 1),
    "neg_inf_entries": None,
  }
  m = r(1, Hq, Nq, Nkv)
  m[0, 0, 3, :] = float("-inf")          # a fully masked row: NaN like SDPA
  m[0, :, 7, Nkv // 2 :] = float("-inf")
  m[0, -1, 11, 1:] = float("-inf")       # a row that sees one key
  cases["neg_inf_entries"] = m
  return cases


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D", [128, 192, 256, 320, 384, 512, 640, 1024])
def test_additive_biases_from_every_source(hip, dtype, D):
  """Additive biases enter this kernel through the S^T accumulators (bias / scale) from three sources — the fp32 row cache in LDS, the
  LDS-DMA staged tiles, element loads; FFPA_FLAG_NO_BIAS_LDS forces the last one.  All three must give the SAME bits (the same
  fp32 values reach the same accumulators), match the oracle (which adds the bias to the scaled score, prefill.cuh:556-658) within
  output rounding, and reproduce NaN rows / single-key rows exactly."""
  B, Hq, Hkv = 2, 4, 2
  for Nq, Nkv, causal in ((130, 333, False), (257, 512, True), (64, 1000, False)):
    q, k, v = _rand((B, Hq, Nq, D), dtype, seed=Nq), _rand((B, Hkv, Nkv, D), dtype, seed=Nkv + 1), _rand((B, Hkv, Nkv, D), dtype, seed=Nkv + 2)
    gen = torch.Generator(device="cuda").manual_seed(D + Nq)
    for name, bias in _bias_cases(B, Hq, Nq, Nkv, dtype, gen).items():
      o, lse = hip.forward(q, k, v, bias, causal, D ** -0.5, kv_bounds=False, num_splits=1)
      og, lg = hip.forward(q, k, v, bias, causal, D ** -0.5, kv_bounds=False, num_splits=1, flags=hip.FLAG_NO_BIAS_LDS)
      tag = f"{name} D{D} {Nq}x{Nkv} causal={causal} {dtype}"
      assert torch.equal(torch.nan_to_num(o.float(), nan=7.0), torch.nan_to_num(og.float(), nan=7.0)) and torch.equal(lse, lg), tag
      _check_vs_oracle(o, lse, q, k, v, causal=causal, bias=_f32(bias), block_keys=32 if D > 512 else 64, name=tag)
      if name == "neg_inf_entries":
        assert torch.isnan(o[0, 0, 3]).all() and torch.equal(o[0, -1, 11], v[0, -1, 0]), tag
      if _has_twin(q) and name in ("dense", "key_bias", "neg_inf_entries"):
        o32, l32 = _twin(hip, q, k, v, bias, causal, kv_bounds=False)
        _same_up_to_rounding(o, lse, o32, l32, dtype, tag)


def test_additive_mask_ranges_and_long_key_biases(hip):
  """kv_bounds with an additive mask: tiles in the neutral interior start from zero accumulators and stage nothing, tiles outside the
  visible range are skipped — the same bits as without the ranges.  A key bias too long for the LDS row cache goes through the staged
  tiles (row stride 0): the same bits as the element loads."""
  Nq, Nkv = 900, 2048
  q, k, v = _rand((1, 4, Nq, D), seed=701), _rand((1, 2, Nkv, D), seed=702), _rand((1, 2, Nkv, D), seed=703)
  rows, cols = torch.arange(Nq, device="cuda")[:, None], torch.arange(Nkv, device="cuda")[None, :]
  for name, keep in (("causal", cols <= rows + 600), ("window", (cols <= rows + 700) & (cols + 200 >= rows)), ("padding", (cols < 1500).expand(Nq, Nkv))):
    for dt in (torch.bfloat16, torch.float32):
      m = torch.zeros(1, 1, Nq, Nkv, dtype=dt, device="cuda").masked_fill(~keep, float("-inf"))
      a, la = hip.forward(q, k, v, m, False, D ** -0.5, kv_bounds=False)
      b, lb = hip.forward(q, k, v, m, False, D ** -0.5, kv_bounds=True)
      assert torch.equal(a, b) and torch.equal(la, lb), (name, dt)
      c, lc = hip.forward(q, k, v, keep.view(1, 1, Nq, Nkv).contiguous(), False, D ** -0.5, kv_bounds=False)
      assert torch.equal(a, c) and torch.equal(la, lc), (name, dt)  # 0 / -inf additive == boolean: the same scores
  Nkv = 20000  # 80 KB of fp32 row cache: does not fit next to the K / V tiles
  q, k, v = _rand((1, 2, 200, D), seed=711), _rand((1, 2, Nkv, D), seed=712), _rand((1, 2, Nkv, D), seed=713)
  kb = _rand((1, 2, 1, Nkv), seed=714)
  a, la = hip.forward(q, k, v, kb, False, D ** -0.5, num_splits=1)
  b, lb = hip.forward(q, k, v, kb, False, D ** -0.5, num_splits=1, flags=hip.FLAG_NO_BIAS_LDS)
  assert torch.equal(a, b) and torch.equal(la, lb)
  s = (q.float() @ k.float().transpose(-1, -2)) * D ** -0.5 + kb.float()
  assert (a.float() - torch.softmax(s, -1) @ v.float()).abs().max().item() < 1e-2
  # in between: a 16-bit key row that does not fit the LDS as fp32 / scale but does as its own 16-bit elements (D = 1024: 16 KiB to spare,
  # 8192 keys) stays on the lean key-bias build, converted at the top of every step — the same bits as the element loads
  D2, Nkv = 1024, 8192
  q, k, v = _rand((1, 2, 200, D2), seed=721), _rand((1, 2, Nkv, D2), seed=722), _rand((1, 2, Nkv, D2), seed=723)
  for dt in (torch.bfloat16, torch.float16):
    kb = (_rand((1, 2, 1, Nkv), seed=724) * 0.5).to(dt)
    kb[..., 100:200] = float("-inf")
    plan = {}
    a, la = hip.forward(q.to(dt), k.to(dt), v.to(dt), kb, False, D2 ** -0.5, num_splits=1, plan_out=plan)
    assert "MK=3" in plan["kernel"], plan
    b, lb = hip.forward(q.to(dt), k.to(dt), v.to(dt), kb, False, D2 ** -0.5, num_splits=1, flags=hip.FLAG_NO_BIAS_LDS, plan_out=plan)
    assert "MK=1" in plan["kernel"] and torch.equal(a, b) and torch.equal(la, lb), dt


@pytest.mark.parametrize("D", [128, 320, 512, 576, 640, 768, 1024])
@pytest.mark.parametrize("case", ["plain", "ragged", "causal_gqa", "bool_mask", "key_bias", "dense_bias", "dropout", "token_major", "split_kv"])
def test_l2_prefetch_changes_no_bit(hip, D, case):
  """The launches of the split-D tiles (D > 512) touch the K/V tile two steps ahead (FwdArgs.l2_prefetch: one dword per line, loaded into
  a register nobody reads): with the touches forced on, forced off and left to the library, every build gives the same bits — outputs
  and LSE — including the steps whose look-ahead tile starts past the last key or past the caller's head dim.  (The D <= 512 builds do
  not carry the touches: there the flags must change nothing either.)"""
  B, Hq, Hkv, Nq, Nkv, causal, kw = 1, 4, 4, 300, 1100, False, {}
  bias = None
  if case == "ragged":
    Nq, Nkv = 129, 97 + 64 * 5
  elif case == "causal_gqa":
    Hkv, causal, Nq, Nkv = 2, True, 512, 777
  elif case == "bool_mask":
    bias = (torch.rand(1, 1, Nq, Nkv, device="cuda") > 0.3)
  elif case == "key_bias":
    bias = (torch.randn(1, 1, 1, Nkv, device="cuda") * 0.5).to(torch.bfloat16)
  elif case == "dense_bias":
    bias = (torch.randn(1, Hq, Nq, Nkv, device="cuda") * 0.5).to(torch.bfloat16)
  elif case == "dropout":
    kw = dict(dropout_p=0.2, philox_seed=11)
  elif case == "split_kv":
    Nq, Nkv, kw = 128, 4096, dict(num_splits=4)
  q, k, v = _rand((B, Hq, Nq, D), seed=5), _rand((B, Hkv, Nkv, D), seed=6), _rand((B, Hkv, Nkv, D), seed=7)
  if case == "token_major":  # [B, N, H, D] storage seen as [B, H, N, D]: rows of one head are H * D elements apart
    k = _rand((B, Nkv, Hkv, D), seed=6).transpose(1, 2)
    v = _rand((B, Nkv, Hkv, D), seed=7).transpose(1, 2)
  if case == "ragged" and D > 8:  # a caller's head dim below the built one: the look-ahead touches stay inside the valid bytes of a row
    q, k, v = q[..., : D - 8], k[..., : D - 8], v[..., : D - 8]
  scale = q.size(-1) ** -0.5
  o_on, l_on = hip.forward(q, k, v, bias, causal, scale, flags=hip.FLAG_L2_PREFETCH, **kw)
  o_off, l_off = hip.forward(q, k, v, bias, causal, scale, flags=hip.FLAG_NO_L2_PREFETCH, **kw)
  o_def, l_def = hip.forward(q, k, v, bias, causal, scale, **kw)
  for o, l in ((o_off, l_off), (o_def, l_def)):
    assert torch.equal(o_on.view(torch.int16), o.view(torch.int16)), f"D={D} {case}: outputs differ"
    assert torch.equal(torch.nan_to_num(l_on, nan=-7.0), torch.nan_to_num(l, nan=-7.0)), f"D={D} {case}: LSE differs"


@pytest.mark.parametrize("case", [(1, 3, 3, 1000, 700, 512, False), (2, 5, 5, 333, 2049, 320, True), (1, 4, 2, 640, 900, 1024, False), (1, 7, 7, 129, 300, 128, True),
                                  (1, 2, 2, 64, 4096, 512, False), (1, 9, 3, 4100, 260, 64, False)])
def test_the_xcd_grouping_of_workgroups_changes_no_bit(hip, case):
  """Which XCD runs which (head, row tile, split) is a launch-side choice (FwdArgs.xcd_group: 1, 2, 4 or 8 XCDs share a head's row tiles, or no
  remapping at all): every choice walks every workgroup of the launch exactly once — workgroup counts that are not multiples of 8, causal
  launches (reversed row tiles), KV-split launches, both kernels — so outputs and LSE agree to the bit."""
  B, Hq, Hkv, Nq, Nkv, D, causal = case
  q, k, v = _rand((B, Hq, Nq, D), seed=21), _rand((B, Hkv, Nkv, D), seed=22), _rand((B, Hkv, Nkv, D), seed=23)
  ref_o, ref_l = hip.forward(q, k, v, None, causal, D ** -0.5, flags=hip.FLAG_NO_XCD_REMAP)
  for flags in [hip.FLAG_XCD_GROUP(g) for g in (1, 2, 4, 8)] + [0]:
    o, l = hip.forward(q, k, v, None, causal, D ** -0.5, flags=flags)
    assert torch.equal(o.view(torch.int16), ref_o.view(torch.int16)), f"{case} flags {flags:#x}: outputs differ"
    assert torch.equal(torch.nan_to_num(l, nan=-7.0), torch.nan_to_num(ref_l, nan=-7.0)), f"{case} flags {flags:#x}: LSE differs"


@pytest.mark.parametrize("shape", [
  # (B, Hq, Hkv, Nq, Nkv, D, causal_offset or None = tail-aligned)
  (1, 4, 4, 1024, 1024, 512, None),     # 8 row tiles: four pairs
  (1, 4, 2, 1152, 1152, 512, None),     # 9 row tiles: the middle one is its own partner; GQA
  (2, 2, 2, 1000, 1500, 512, None),     # ragged rows and keys, tail-aligned against a longer context
  (1, 4, 4, 1024, 2048, 512, 0),        # top-left causal against a longer context
  (1, 2, 2, 640, 640, 1024, None),      # split-D tiles (64 rows per tile, the softmax pipeline): ten tiles
  (1, 2, 2, 448, 777, 1024, None),      # ... seven tiles, ragged keys
  (1, 4, 4, 1024, 1024, 320, None),     # 128-key tiles
  (1, 4, 1, 512, 512, 128, None),
  (1, 2, 2, 384, 384, 576, None),       # D % 128 == 64: the un-pipelined split-D loop
])
def test_paired_row_tiles_are_bit_identical_to_unpaired(hip, shape):
  """FFPA_FLAG_PAIR_TILES (round 6): under the causal flag workgroup i of a head walks row tile n - 1 - i and then row tile i.  Per row nothing changes — the
  same tile, the same recurrence — so O and LSE must equal the unpaired launch bit for bit, with dropout too (the Philox counter is the row's)."""
  B, Hq, Hkv, Nq, Nkv, D, off = shape
  q, k, v = _rand((B, Hq, Nq, D), seed=7), _rand((B, Hkv, Nkv, D), seed=8), _rand((B, Hkv, Nkv, D), seed=9)
  for drop in (0.0, 0.15):
    kw = dict(causal_offset=off, dropout_p=drop, philox_seed=1234, philox_offset=8)
    o0, l0 = hip.forward(q, k, v, None, True, D ** -0.5, flags=hip.FLAG_NO_PAIR_TILES, **kw)
    plan = {}
    o1, l1 = hip.forward(q, k, v, None, True, D ** -0.5, flags=hip.FLAG_PAIR_TILES, plan_out=plan, **kw)
    assert torch.equal(o0, o1) and torch.equal(l0, l1), (shape, drop, (o0.float() - o1.float()).abs().max().item())
  # and against SDPA (top-left / tail-aligned mask built explicitly)
  rows = torch.arange(Nq, device=q.device)[:, None]
  cols = torch.arange(Nkv, device=q.device)[None, :]
  keep = cols <= rows + ((Nkv - Nq) if off is None else off)
  ref = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=keep, enable_gqa=Hq != Hkv)
  o1, _ = hip.forward(q, k, v, None, True, D ** -0.5, causal_offset=off, flags=hip.FLAG_PAIR_TILES)
  assert (o1.float() - ref.float()).abs().max().item() <= 2e-2


@pytest.mark.parametrize("shape", [
  dict(B=1, Hq=32, Hkv=8, Nq=2048, Nkv=2048, D=512, dtype=torch.bfloat16),
  dict(B=3, Hq=16, Hkv=2, Nq=777, Nkv=777, D=320, dtype=torch.float16),       # chunks of 2 (groups of 8 in eight chunks), ragged tail, odd batch
  dict(B=1, Hq=64, Hkv=8, Nq=640, Nkv=1900, D=1024, dtype=torch.bfloat16),     # tail-aligned against a longer context, split-D tiles
  dict(B=2, Hq=32, Hkv=8, Nq=1024, Nkv=1024, D=128, dtype=torch.bfloat16),
])
def test_head_chunk_order_computes_the_same_bits(shape):
  """Causal GQA launches run in the head-chunk workgroup order (the packed-sequence kernel's, in its dense mode: ffpa_capi.hip::pick_dense_head_chunk): the
  same tile text on the same arguments — LSE (both dtypes) and O (bf16) bit-identical to the (batch, head, row tile) order (FFPA_FLAG_NO_HEAD_CHUNKS), fp16 O up
  to the compiler's last-instruction rounding choice; the plan says which ran."""
  from ffpa_attn_amd import hip

  B, Hq, Hkv, Nq, Nkv, D, dtype = (shape[k] for k in ("B", "Hq", "Hkv", "Nq", "Nkv", "D", "dtype"))
  g = torch.Generator(device="cuda").manual_seed(Nq + D)
  q = torch.randn((B, Hq, Nq, D), dtype=dtype, device="cuda", generator=g)
  k = torch.randn((B, Hkv, Nkv, D), dtype=dtype, device="cuda", generator=g)
  v = torch.randn((B, Hkv, Nkv, D), dtype=dtype, device="cuda", generator=g)
  plan, plan_off = {}, {}
  o, lse = hip.forward(q, k, v, None, True, D ** -0.5, plan_out=plan, num_splits=1)
  o2, lse2 = hip.forward(q, k, v, None, True, D ** -0.5, flags=hip.FLAG_NO_HEAD_CHUNKS | hip.FLAG_NO_PAIR_TILES, plan_out=plan_off, num_splits=1)
  assert "head chunks of" in plan["kernel"] and "varlen" not in plan_off["kernel"], (plan, plan_off)
  assert torch.equal(lse, lse2)
  if dtype == torch.bfloat16:
    assert torch.equal(o, o2)
  else:
    # fp16: the same accumulators; the very last instruction (O / l rounded to fp16) is hipcc's per-element choice between v_fma_mixlo_f16 (one rounding) and
    # v_mul_f32 + v_cvt (two), made differently in the two kernels: ~ 2^-13 of the elements differ by one fp16 spacing (tests/test_varlen_gpu.py)
    diff = (o.float() - o2.float()).abs()
    assert torch.all(diff <= torch.maximum(o.float().abs(), o2.float().abs()).clamp_min(2.0 ** -14) * 2.0 ** -10) and (diff > 0).float().mean().item() <= 2e-3
  # strided inputs ([B, N, H, D] storage) through the same route
  qs, ks, vs = (t.transpose(1, 2).contiguous().transpose(1, 2) for t in (q, k, v))
  o3, lse3 = hip.forward(qs, ks, vs, None, True, D ** -0.5, num_splits=1)
  assert torch.equal(o, o3) and torch.equal(lse, lse3)


@pytest.mark.parametrize("shape", [
  dict(B=1, Hq=8, Hkv=8, Nq=4096, Nkv=4096, D=512, dtype=torch.bfloat16, ranges=0),    # the rule's own: 256 row tiles = one round on 256 CUs
  dict(B=2, Hq=3, Hkv=1, Nq=1500, Nkv=1900, D=320, dtype=torch.float16, ranges=3),     # forced: batch, GQA, ragged last row tile, tail-aligned against a longer context
  dict(B=1, Hq=4, Hkv=4, Nq=1000, Nkv=1000, D=1024, dtype=torch.bfloat16, ranges=5),   # split-D tiles, more ranges than the first row tiles have KV tiles
  dict(B=1, Hq=16, Hkv=2, Nq=640, Nkv=640, D=128, dtype=torch.bfloat16, ranges=2),     # head chunks of 2 AND ranges
])
def test_causal_kv_ranges_per_row_tile(shape):
  """Causal launches of one round of workgroups split every row tile's OWN visible KV tiles (ffpa_capi.hip::pick_tile_ranges; the packed-sequence kernel in its
  dense mode computes the ranges on the device, the dense merge kernel combines the fp32 partials): the rule's shape and forced counts (FLAG_TILE_RANGES |
  FLAG_FORCE_SPLITS) against the one-range launch (to merge rounding), the oracle on a row subset and SDPA; strided views; the opt-outs."""
  from ffpa_attn_amd import hip

  B, Hq, Hkv, Nq, Nkv, D, dtype, ranges = (shape[k] for k in ("B", "Hq", "Hkv", "Nq", "Nkv", "D", "dtype", "ranges"))
  if ranges == 0 and hip.load_library().ffpa_attn_query(8) != 256:  # (FFPA_QUERY_DEVICE_CUS: what the plan prices, FFPA_HIP_FAKE_CUS included)
    pytest.skip("the rule's shape is one round of workgroups on 256 CUs")
  g = torch.Generator(device="cuda").manual_seed(Nq + D)
  q = torch.randn((B, Hq, Nq, D), dtype=dtype, device="cuda", generator=g)
  k = torch.randn((B, Hkv, Nkv, D), dtype=dtype, device="cuda", generator=g)
  v = torch.randn((B, Hkv, Nkv, D), dtype=dtype, device="cuda", generator=g)
  kw = dict(num_splits=ranges, flags=hip.FLAG_TILE_RANGES | hip.FLAG_FORCE_SPLITS) if ranges else {}
  plan, plan1 = {}, {}
  o, lse = hip.forward(q, k, v, None, True, D ** -0.5, plan_out=plan, **kw)
  o1, lse1 = hip.forward(q, k, v, None, True, D ** -0.5, num_splits=1, plan_out=plan1)
  assert "KV ranges per row tile" in plan["kernel"] and plan["kernel"].endswith("+ ffpa_fwd_merge_kernel") and plan["splits"] == (ranges or 2), plan
  assert plan1["splits"] == 1 and "KV ranges" not in plan1["kernel"], plan1
  # merge rounding (tests/test_varlen_gpu.py::_same_to_merge_rounding): every range rounds its P entries against its own running max — per element one output
  # spacing + 2^-7 (bf16) / 2^-10 (fp16) of the row's largest |O| (the first rows of a causal launch average a handful of V rows: |O| ~ 1 ... 3)
  a, b = o.float(), o1.float()
  eps = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
  allow = torch.maximum(a.abs(), b.abs()).clamp_min(2.0 ** -14) * eps + b.abs().amax(dim=-1, keepdim=True) * eps
  diff = (a - b).abs()
  assert torch.all(diff <= allow) and (diff / allow).mean().item() < 0.1, f"O differs from the one-range launch by {(diff / allow).max().item():.2f} x the allowance"
  assert (lse - lse1).abs().max().item() <= 1e-4
  rr, cc = torch.arange(Nq, device="cuda")[:, None], torch.arange(Nkv, device="cuda")[None, :]
  ref = F.scaled_dot_product_attention(q, k, v, attn_mask=cc <= rr + (Nkv - Nq), enable_gqa=Hq != Hkv)
  _close(o, ref, dtype, f"tile ranges vs SDPA {shape}")
  # the oracle on whole row tiles of the first and the last quarter of one head (their ranges differ most)
  for r0 in (0, (Nq - 1) // 128 * 128 - 128):
    r0 = max(r0, 0)
    _check_vs_oracle(o[:1, :1], lse[:1, :1], q[:1, :1], k[:1, :1], v[:1, :1], causal=True, rows=(r0, min(Nq, r0 + 256)), block_keys=plan["block_keys"],
                     name=f"tile ranges vs oracle rows {r0} {shape}", split=True)
  # strided inputs ([B, N, H, D] storage): the same plan, the same bits
  qs, ks, vs = (t.transpose(1, 2).contiguous().transpose(1, 2) for t in (q, k, v))
  o3, lse3 = hip.forward(qs, ks, vs, None, True, D ** -0.5, **kw)
  assert torch.equal(o, o3) and torch.equal(lse, lse3)
  # the opt-outs keep one range: the flag, the deterministic flag
  for flag in (hip.FLAG_NO_TILE_RANGES, hip.FLAG_DETERMINISTIC):
    p2 = {}
    o4, lse4 = hip.forward(q, k, v, None, True, D ** -0.5, flags=flag, plan_out=p2)
    assert "KV ranges" not in p2["kernel"] and (p2["splits"] == 1 or flag == hip.FLAG_NO_TILE_RANGES), p2
    if p2["splits"] == 1:
      assert torch.equal(lse4, lse1) and (dtype != torch.bfloat16 or torch.equal(o4, o1))
