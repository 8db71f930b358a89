"""The wide-row prefill tile (csrc/ffpa_fwd_m16w_kernel.h; `pytest -m gpu`): 16 RH query rows per wave with RH sized to the accumulator
file (D = 320: 48 rows per wave, 192 per workgroup), 64-key tiles double-buffered in the LDS, one workgroup barrier per KV step.

Forced here with FLAG_WIDE_TILE (the launch side takes it on its own where a launch's rounds come out cheaper: test_capi.py pins that rule).
Pinned: the oracle (the reference's recurrence restated on the CPU) with the kernel's own key blocking, the 32-row tile of the same head dim
up to output rounding (another summation order of the same sums), exact NaN / -inf patterns, and what must hold to the bit inside one
build: determinism, head independence, mask ranges on / off, strided views, batch / head placement.
"""

import pytest
import torch

from test_fwd_gpu import _check_vs_oracle, _rand, _within_north_star, hip  # noqa: F401  (fixture + helpers)
from test_m16_gpu import _same_up_to_rounding

pytestmark = pytest.mark.gpu

WIDE_DIMS = {320: 192}  # head dim of the library object -> rows per workgroup


def _wide(hip, q, k, v, bias=None, causal=False, scale=None, **kw):
  plan = {}
  kw.setdefault("num_splits", 1)
  o, lse = hip.forward(q, k, v, bias, causal, q.size(-1) ** -0.5 if scale is None else scale, flags=hip.FLAG_WIDE_TILE | kw.pop("flags", 0), plan_out=plan, **kw)
  dk = (q.size(-1) + 63) // 64 * 64
  assert plan["kernel"].startswith(f"ffpa_fwd_m16w_kernel<{'bf16' if q.dtype == torch.bfloat16 else 'fp16'}, {dk}, RH={WIDE_DIMS[dk] // 64}"), plan
  assert plan["block_rows"] == WIDE_DIMS[dk] and plan["block_keys"] == 64, plan
  return o, lse


def _narrow(hip, q, k, v, bias=None, causal=False, scale=None, **kw):
  plan = {}
  kw.setdefault("num_splits", 1)
  o, lse = hip.forward(q, k, v, bias, causal, q.size(-1) ** -0.5 if scale is None else scale, flags=hip.FLAG_NO_WIDE_TILE | kw.pop("flags", 0), plan_out=plan, **kw)
  assert plan["kernel"].startswith("ffpa_fwd_m16_kernel<"), plan
  return o, lse


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D", sorted(WIDE_DIMS))
@pytest.mark.parametrize("case", [(1, 2, 2, 128, 64, False), (1, 2, 1, 200, 333, False), (2, 4, 2, 384, 384, True), (1, 2, 2, 77, 1000, True),
                                  (1, 1, 1, 640, 1500, False), (1, 4, 4, 33, 65, True), (2, 2, 1, 129, 63, False), (1, 2, 2, 192, 128, False),
                                  (1, 1, 1, 193, 129, True), (1, 3, 1, 575, 2048, False), (1, 2, 2, 1000, 1000, True)])
def test_matches_the_oracle_and_the_32_row_tile(hip, dtype, D, case):
  B, Hq, Hkv, Nq, Nkv, causal = case
  q, k, v = _rand((B, Hq, Nq, D), dtype, seed=Nq), _rand((B, Hkv, Nkv, D), dtype, seed=Nkv + 1), _rand((B, Hkv, Nkv, D), dtype, seed=Nkv + 2)
  ow, lw = _wide(hip, q, k, v, None, causal)
  on, ln = _narrow(hip, q, k, v, None, causal)
  _same_up_to_rounding(ow, lw, on, ln, dtype, str(case))
  _check_vs_oracle(ow, lw, q, k, v, causal=causal, block_keys=64, name=f"m16w D{D} {case}")
  ow2, lw2 = _wide(hip, q, k, v, None, causal)
  assert torch.equal(torch.nan_to_num(ow, nan=7.0), torch.nan_to_num(ow2, nan=7.0)) and torch.equal(torch.nan_to_num(lw, nan=7.0), torch.nan_to_num(lw2, nan=7.0))  # deterministic


@pytest.mark.parametrize("D", sorted(WIDE_DIMS))
def test_causal_offsets_tails_and_fully_masked_rows(hip, D):
  Nq, Nkv = 300, 700
  q, k, v = _rand((1, 2, Nq, D), seed=11), _rand((1, 2, Nkv, D), seed=12), _rand((1, 2, Nkv, D), seed=13)
  for off in (0, 400, -40, 650):  # SDPA-style, tail-aligned, rows with no visible key (NaN), almost everything visible
    ow, lw = _wide(hip, q, k, v, None, True, causal_offset=off)
    on, ln = _narrow(hip, q, k, v, None, True, causal_offset=off)
    _same_up_to_rounding(ow, lw, on, ln, q.dtype, f"offset {off}")
    _check_vs_oracle(ow, lw, q, k, v, causal=True, causal_offset=off, block_keys=64, name=f"m16w offset {off}")
  o, lse = _wide(hip, q, k, v, None, True, causal_offset=-40)
  assert torch.isnan(o[:, :, :40]).all() and not torch.isnan(o[:, :, 40:]).any()
  assert torch.equal(o[0, :, 40], v[0, :, 0])  # a row that sees exactly one key returns that key's V row
  assert torch.isneginf(lse[:, :, :40]).all() or torch.isnan(lse[:, :, :40]).all()


@pytest.mark.parametrize("D", sorted(WIDE_DIMS))
def test_boolean_masks_vector_and_byte_paths_and_ranges(hip, D):
  """Mask bytes are read 4 at a time (unit key stride, 16-byte aligned rows, full tile) or one by one; mask ranges (per 32-row block; a wave's
  48 rows straddle two of them) skip tiles and mask reads: with and without the ranges the SAME bits."""
  for Nq, Nkv in ((513, 1024), (513, 1000), (130, 777), (400, 2048)):
    q, k, v = _rand((1, 2, Nq, D), seed=61), _rand((1, 2, Nkv, D), seed=62), _rand((1, 2, Nkv, D), seed=63)
    g = torch.Generator(device="cuda").manual_seed(Nq + Nkv)
    mask = torch.rand(1, 2, Nq, Nkv, device="cuda", generator=g) > 0.3
    mask[0, 0, 5, :] = False
    mask[0, :, 9, :128] = False
    mask[0, 1, 100, 1:] = False
    mask[0, 1, 100, 0] = True
    ow, lw = _wide(hip, q, k, v, mask, False, kv_bounds=False)
    on, ln = _narrow(hip, q, k, v, mask, False, kv_bounds=False)
    _same_up_to_rounding(ow, lw, on, ln, q.dtype, f"bool {Nq}x{Nkv}")
    assert torch.isnan(ow[0, 0, 5]).all() and torch.equal(ow[0, 1, 100], v[0, 1, 0])
    bias = torch.zeros(mask.shape, dtype=torch.float32, device="cuda").masked_fill(~mask, float("-inf"))
    _check_vs_oracle(ow, lw, q, k, v, bias=bias.cpu().numpy(), block_keys=64, name=f"m16w bool {Nq}x{Nkv}")
    for m in (mask, mask[:, :1], mask[:, :, :1], mask[..., ::2].repeat_interleave(2, -1)[..., :Nkv]):  # broadcast heads / rows, odd strides
      a, la = _wide(hip, q, k, v, m, False, kv_bounds=False)
      b_, lb = _wide(hip, q, k, v, m, False, kv_bounds=True)
      assert torch.equal(torch.nan_to_num(a, nan=7.0), torch.nan_to_num(b_, nan=7.0)) and torch.equal(torch.nan_to_num(la, nan=7.0), torch.nan_to_num(lb, nan=7.0))
      c_, lc = _narrow(hip, q, k, v, m, False, kv_bounds=True)
      _same_up_to_rounding(a, la, c_, lc, q.dtype, "broadcast / strided mask")


@pytest.mark.parametrize("D", sorted(WIDE_DIMS))
def test_explicit_causal_masks_equal_the_causal_flag(hip, D):
  """BASELINE config 4 in small: GQA cross attention with SDPA's top-left causal mask, as the flag (causal_offset = 0) and as an explicit boolean
  mask with and without ranges — the same visible keys, the same 64-key blocking: the same bits."""
  B, Hq, Hkv, Nq, Nkv = 2, 8, 2, 1100, 512
  q, k, v = _rand((B, Hq, Nq, D), seed=5), _rand((B, Hkv, Nkv, D), seed=6), _rand((B, Hkv, Nkv, D), seed=7)
  mask = torch.ones(Nq, Nkv, dtype=torch.bool, device="cuda").tril().view(1, 1, Nq, Nkv)
  of, lf = _wide(hip, q, k, v, None, True, causal_offset=0)
  for kvb in (False, True):
    om, lm = _wide(hip, q, k, v, mask, False, kv_bounds=kvb)
    assert torch.equal(of, om) and torch.equal(lf, lm), f"kv_bounds={kvb}"
  _check_vs_oracle(of, lf, q, k, v, causal=True, causal_offset=0, block_keys=64, rows=(0, 600), name="m16w config-4-like")


@pytest.mark.parametrize("D", [264, 296, 312, 320])
def test_head_dims_between_the_built_multiples_of_64(hip, D):
  q, k, v = _rand((1, 3, 250, D), seed=21), _rand((1, 3, 500, D), seed=22), _rand((1, 3, 500, D), seed=23)
  ow, lw = _wide(hip, q, k, v, None, False)
  _check_vs_oracle(ow, lw, q, k, v, block_keys=64, name=f"m16w D{D}")
  oc, lc = _wide(hip, q, k, v, None, True)
  _check_vs_oracle(oc, lc, q, k, v, causal=True, block_keys=64, name=f"m16w D{D} causal")


@pytest.mark.parametrize("D", sorted(WIDE_DIMS))
def test_strided_views_scales_and_placement(hip, D):
  B, H, N = 2, 4, 300
  qkv = _rand((B, N, 3, H, D), seed=31)  # token-major packed projection output: [B, N, 3, H, D] -> three [B, H, N, D] views
  q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
  ow, lw = _wide(hip, q, k, v, None, True)
  oc, lc = _wide(hip, q.contiguous(), k.contiguous(), v.contiguous(), None, True)
  assert torch.equal(ow, oc) and torch.equal(lw, lc)
  # a (batch, head) slice does not depend on what else is in the launch
  o1, l1 = _wide(hip, q[1:, 2:3].contiguous(), k[1:, 2:3].contiguous(), v[1:, 2:3].contiguous(), None, True)
  assert torch.equal(o1, oc[1:, 2:3]) and torch.equal(l1, lc[1:, 2:3])
  # XCD placement is speed only
  o2, l2 = _wide(hip, q, k, v, None, True, flags=hip.FLAG_NO_XCD_REMAP)
  assert torch.equal(o2, oc) and torch.equal(l2, lc)
  for scale in (0.0, -0.05, 0.1):  # (a peaky softmax — scale 0.3: |O| up to 3, p / l ~ 1 — moves whole output ulps between two key blockings: the oracle tests cover it)
    os_, ls_ = _wide(hip, q, k, v, None, False, scale=scale)
    on, ln = _narrow(hip, q, k, v, None, False, scale=scale)
    _same_up_to_rounding(os_, ls_, on, ln, q.dtype, f"scale {scale}")
  # exact recurrence (threshold 0) and the lazy one agree up to rounding; rows whose max grows by more than the threshold take the rescale path
  big = q.clone()
  big[:, :, :, :8] *= 6.0
  oa, la = _wide(hip, big, k, v, None, False, rescale_threshold=0.0)
  ob, lb = _wide(hip, big, k, v, None, False)
  _same_up_to_rounding(oa, la, ob, lb, q.dtype, "lazy vs exact rescale")
  _check_vs_oracle(ob, lb, big, k, v, block_keys=64, name="m16w rescale path")


@pytest.mark.parametrize("D", sorted(WIDE_DIMS))
def test_kv_splits_merge_to_the_unsplit_result(hip, D):
  q, k, v = _rand((1, 2, 400, D), seed=41), _rand((1, 2, 4096, D), seed=42), _rand((1, 2, 4096, D), seed=43)
  o1, l1 = _wide(hip, q, k, v, None, False)
  plan = {}
  o3, l3 = hip.forward(q, k, v, None, False, D ** -0.5, flags=hip.FLAG_WIDE_TILE | hip.FLAG_FORCE_SPLITS, num_splits=3, plan_out=plan)
  assert plan["splits"] == 3 and plan["kernel"].startswith("ffpa_fwd_m16w_kernel<") and plan["kernel"].endswith("+ ffpa_fwd_merge_kernel"), plan
  _same_up_to_rounding(o3, l3, o1, l1, q.dtype, "3 KV splits")
  _check_vs_oracle(o3, l3, q, k, v, block_keys=64, split=True, name="m16w 3 KV splits")


def test_config4_shape_at_full_size(hip):
  """BASELINE config 4 (B2 Hq32 Hkv8 Nq8192 Nkv2048 D320, SDPA's top-left causal mask as an explicit boolean tensor): the wide tile against the
  32-row tile on every row, against the oracle on 512 rows, against SDPA within the north star's 1e-2."""
  import torch.nn.functional as F

  B, Hq, Hkv, Nq, Nkv, D = 2, 32, 8, 8192, 2048, 320
  torch.manual_seed(0)
  q = torch.randn(B, Hq, Nq, D, dtype=torch.bfloat16, device="cuda")
  k = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  v = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  mask = torch.ones(Nq, Nkv, dtype=torch.bool, device="cuda").tril().view(1, 1, Nq, Nkv)
  ow, lw = _wide(hip, q, k, v, mask, False)
  on, ln = _narrow(hip, q, k, v, mask, False)
  _same_up_to_rounding(ow, lw, on, ln, q.dtype, "config 4")
  for i, head in enumerate((0, 3, 8, 13, 18, 23, 28, 31)):  # 8 heads x 64 rows, both batch elements
    r0 = (0, 500, 1100, 2040, 3000, 4500, 6000, 8128)[i]
    sl, kv, bb = slice(head, head + 1), slice(head // 4, head // 4 + 1), slice(i % 2, i % 2 + 1)
    _check_vs_oracle(ow[bb, sl], lw[bb, sl], q[bb, sl], k[bb, kv], v[bb, kv], causal=True, causal_offset=0, rows=(r0, r0 + 64), block_keys=64, name=f"cfg4 wide h{head}")
  ref = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, enable_gqa=True)
  _within_north_star(ow, ref)
