"""GPU parity tests proper (`pytest -m gpu`, one MI355X): the HIP kernel, called through the C-ABI,
against the pinned CPU oracle, the committed SDPA fixtures, and PyTorch SDPA on the same device.

Mirrors the coverage of the reference's tests/test_ffpa_fwd.py (shapes :32-45, boundary seqlens
:1086-1104, cross-attention tails :1110-1143, GQA/MQA :1158-1196, causal self / tail-aligned cross
:1226-1304, masks :291-336, decode Nq in {1,7} :929-958) with its tolerance (atol = rtol = 2e-2 bf16 /
1e-2 fp16, :106-113) plus the tighter max-abs <= 1e-2 the north star states, and adds what a
from-scratch kernel needs: a bit-exact cross-check of the LDS-DMA / transpose-read path against a
register-staged twin, forced lazy-rescale inputs, strided views and size-independent properties at
the BASELINE sizes.
"""

import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, case_bias
from oracle import ffpa_oracle as fo

pytestmark = pytest.mark.gpu

TOL = {torch.bfloat16: 2e-2, torch.float16: 1e-2}  # reference test tolerance (atol = rtol)
NORTH_STAR_MAX_ABS = 1e-2


def _within_north_star(o, ref):
  """max-abs <= 1e-2 vs SDPA — or one bf16 spacing (2^-7 relative) where |O| is so large (early causal
  rows average only a few V rows, |O| ~ 2-4) that a single output ulp already exceeds 1e-2."""
  d = (o.float() - ref.float()).abs()
  lim = torch.maximum(torch.full_like(d, NORTH_STAR_MAX_ABS), ref.float().abs() * 2.0 ** -7)
  assert torch.all(d <= lim), f"max abs {d.max().item():.3e}, worst excess {(d - lim).max().item():.3e}"
  return d.max().item()


@pytest.fixture(scope="module")
def hip():
  if not torch.cuda.is_available():
    pytest.fail("these tests need a GPU; run with -m 'not gpu' on CPU boxes")
  from ffpa_attn_amd import hip as h

  h.load_library()  # fail loudly if the extension is missing: there is no fallback
  return h


def _t(bits, dtype):
  tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
  return torch.from_numpy(bits.view(np.int16).copy()).view(tdt).cuda()


def _f32(t):
  return t.detach().float().cpu().numpy()


def _short_query_keys(hip, D):
  """Keys per tile of the short-query (Nq <= 32) launches: what the oracle's recurrence has to be blocked by to follow the kernel."""
  plan = {}
  q = torch.zeros(1, 1, 1, D, dtype=torch.bfloat16, device="cuda")
  k = torch.zeros(1, 1, 256, D, dtype=torch.bfloat16, device="cuda")
  hip.forward(q, k, k, None, False, 1.0, plan_out=plan, num_splits=1)
  return plan["block_keys"]


def _check_vs_oracle(o_gpu, lse_gpu, q, k, v, *, causal=False, causal_offset=None, bias=None, rows=None,
                     block_keys=64, threshold=8.0, name="", split=False):
  """o_gpu within one storage-dtype rounding of the oracle's unrounded result, LSE to fp32 noise.  ``split``: the launch splits the KV
  axis over workgroups (fp32 partials + LSE merge) — every split rounds its P entries against its OWN running max, so the 16-bit
  rounding noise of P is not shared with the oracle's single walk and enters the allowance (five sigma of both sides)."""
  qb, dt = fo.torch_to_bits(q)
  kb, _ = fo.torch_to_bits(k)
  vb, _ = fo.torch_to_bits(v)
  _, o32, lse, (pmax, p2sum) = fo.oracle_forward(qb, kb, vb, dt, causal=causal, causal_offset=causal_offset, bias=bias, rows=rows,
                                                 block_keys=block_keys, threshold=threshold, return_pmax="both")
  got = _f32(o_gpu)
  r0, r1 = (0, got.shape[2]) if rows is None else rows
  got, want, pmax, p2sum = got[:, :, r0:r1], o32[:, :, r0:r1], pmax[:, :, r0:r1], p2sum[:, :, r0:r1]
  finite = np.isfinite(want)
  assert np.array_equal(np.isnan(got), np.isnan(want)), f"{name}: NaN pattern differs"
  # |round(kernel_f32) - oracle_f32| <= half a storage ulp of the result + the effect of P entries whose
  # fp32 value sits on a rounding boundary and rounds the other way in the kernel (v_exp_f32 / MFMA summation
  # order vs libm / sequential): one flip moves O by ulp * p/l * |v|.  The allowance is that expression with the ROW's largest p/l (the
  # oracle returns it) and the largest |v| — 2.5e-3 at 32 keys (p/l ~ 0.1 ... 1: the constant earlier rounds used everywhere, still the cap),
  # ~ 1e-4 at 8192 keys (p/l ~ 5e-3), where a constant 2.5e-3 would be a quarter of an output's standard deviation and could not see a
  # dropped KV tile (round-3 review; test_oracle_check_catches_a_dropped_kv_tile).
  ulp = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
  vmax = float(v.detach().float().abs().max().item())
  flip_cap = 2.5e-3 if dt == "bf16" else 4e-4
  noise = 0.0
  if split:
    vrms = float(v.detach().float().pow(2).mean().sqrt().item())
    noise = 5.0 * (0.5 * ulp / np.sqrt(3.0)) * np.sqrt(2.0 * np.nan_to_num(p2sum, nan=1.0)) * vrms
  # (3 x: the kernel's scores differ from the oracle's by the fp32 summation order of the MFMA — ~ 1e-5 in the log2 domain, i.e. about one P
  # entry in 500 lands on the other side of a 16-bit rounding boundary: rows of a few hundred keys see a handful of flips among their larger
  # entries; 1.5 x failed 64 of the suite's 1447 cases by up to 3e-4, all at 250 ... 3000 keys)
  # (the cap never cuts below ONE flip of the row's largest entry at the largest |v|: a row of two visible keys whose second P entry sits on a 16-bit rounding
  # midpoint moves by ulp x |v| / l = 7.4e-3 when it rounds the other way — packed fuzz seed 1122, row 1 of a causal sequence: p' = 0.55665 between 0.5547 and
  # 0.5586, v = -2.95, oracle -1.3560, kernel -1.3634 -> bf16 -1.3672, exact -1.3596; tools/visits/dbg_seed1122.py.  Rows of a few hundred keys and more are
  # unaffected: there one flip is worth 1e-4 and the three-flip expression stays under the cap.)
  one_flip = ulp * np.nan_to_num(pmax, nan=1.0) * vmax
  flip = np.minimum(np.maximum(flip_cap, one_flip), np.maximum(3.0 * one_flip, 2e-5) + noise)[..., None] * np.ones_like(want)
  err = np.abs(got - want)[finite]
  half_ulp = (ulp * np.maximum(np.abs(want), 2.0 ** -6))[finite]
  flip = flip[finite]
  assert (err <= half_ulp + flip).all(), f"{name}: max err {err.max():.3e} (worst excess {(err - half_ulp - flip).max():.3e})"
  # flips are rare: the MEAN error must stay at pure output-rounding level
  assert err.mean() <= 0.5 * (half_ulp + np.minimum(3e-4, 2e-5 + flip / 3)).mean(), f"{name}: mean err {err.mean():.3e}"
  if lse_gpu is not None:
    lg, lw = _f32(lse_gpu)[:, :, r0:r1], lse[:, :, r0:r1]
    fin = np.isfinite(lw)
    assert np.array_equal(np.isneginf(lg), np.isneginf(lw)), f"{name}: LSE -inf pattern"
    np.testing.assert_allclose(lg[fin], lw[fin], atol=2e-4, rtol=2e-5, err_msg=name)


def _close(a, b, dtype, name=""):
  tol = TOL[dtype]
  a, b = a.float(), b.float()
  assert torch.equal(torch.isnan(a), torch.isnan(b)), name
  m = ~torch.isnan(b)
  assert torch.all((a - b).abs()[m] <= tol + tol * b.abs()[m]), f"{name}: {(a - b).abs()[m].max().item():.3e}"


def _rand(shape, dtype=torch.bfloat16, seed=0):
  g = torch.Generator(device="cuda").manual_seed(seed)
  return torch.randn(shape, dtype=dtype, device="cuda", generator=g)


# ----------------------------------------------------------------------------- committed fixtures
def test_small_cases_vs_oracle_and_sdpa_fixtures(hip, small_cases):
  for c in small_cases:
    q, k, v = (_t(c[n], c["dtype"]) for n in "qkv")
    bias = case_bias(c)
    bias_t = None
    if bias is not None:
      if "mask_bits" in c:
        bias_t = _t(c["mask_bits"], c["dtype"])
      elif c["mask"].dtype == np.bool_:
        bias_t = torch.from_numpy(bias).to(q.dtype).cuda()  # bool -> 0/-inf in q.dtype (functional.py:891-898)
      else:
        bias_t = torch.from_numpy(bias).cuda()
    o, lse = hip.forward(q, k, v, bias_t, c["causal"], 1.0 / c["D"] ** 0.5, causal_offset=c["causal_offset"])
    bc = hip.tile_config(hip.padded_head_dim(c["D"]))["block_keys"]
    _check_vs_oracle(o, lse, q, k, v, causal=c["causal"], causal_offset=c["causal_offset"], bias=bias, block_keys=bc,
                     name=c["name"])
    want = _t(c["o_sdpa"], c["dtype"])
    _close(o, want, q.dtype, c["name"])
    assert (o.float() - want.float()).abs().max().item() <= 1.2e-2, c["name"]
    np.testing.assert_allclose(_f32(lse), c["lse_f64"], atol=3e-4, rtol=3e-5, err_msg=c["name"])


def test_kernel_matches_the_executed_reference_triton_kernel(hip):
  """fp16 and bf16 outputs of the reference's own Triton forward (tests/golden/make_triton_golden.py, run under TRITON_INTERPRET=1
  in the authoring container) vs the HIP kernel on the re-created inputs: tails, tail-aligned causal + GQA, additive
  biases with -inf entries, late score spikes (the lazy-rescale branch), D = 320 / 512 / 1024."""
  import sys

  from test_oracle import _bits_to_f32, _triton_cases, triton_fixture_limits

  sys.path.insert(0, GOLDEN)
  import triton_cases as mt

  for case, (q, k, v, bias), o_ref_bits, lse_ref in _triton_cases():
    name, Nkv, D, causal, dtype = case[0], case[5], case[6], case[7], case[9]
    tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
    qt, kt, vt = (torch.from_numpy(a.view(np.int16).copy()).view(tdt).cuda() for a in (q, k, v))
    bt = None if bias is None else torch.from_numpy(bias.view(np.int16).copy()).view(tdt).cuda()
    o, lse = hip.forward(qt, kt, vt, bt, causal, D ** -0.5, num_splits=1 if Nkv >= 4096 else 0)
    want = _bits_to_f32(o_ref_bits, dtype)
    # (the rows' largest p / l for the long cases' allowance comes from the oracle: test infrastructure, like the limits themselves)
    pmax = fo.oracle_forward(q.view(np.uint16), k.view(np.uint16), v.view(np.uint16), dtype, causal=causal, bias=mt.to_f32(bias, dtype),
                             return_pmax="both")[3] if Nkv >= 4096 else None
    lim, mean_lim = triton_fixture_limits(want, dtype, Nkv, pmax, float(np.abs(mt.to_f32(v, dtype)).max()))
    d = np.abs(_f32(o) - want)
    # (short rows: one output ulp of the binade above, row scale for the spike cases; BASELINE-length rows: one spacing of the element's own binade)
    assert bool((d <= lim).all()) and d.mean() <= mean_lim * (1.25 if dtype == "fp16" and Nkv < 4096 else 1.0), (name, d.max(), d.mean())
    assert (lse.cpu() - torch.from_numpy(lse_ref)).abs().max().item() <= 3e-5, name


def test_kernel_dropout_matches_the_executed_reference_triton_kernel(hip):
  """Dropout against EXECUTED reference code: the reference's Triton forward run with dropout_p / philox_seed / philox_offset in the authoring container
  (tests/golden/ref_triton_dropout.npz) vs the HIP kernel on the re-created inputs — the Philox stream, its element offsets (inside a quad, batch / head
  terms, 62-bit seed), the keep rule and the 1 / (1 - p) scaling are the reference's, not just the oracle's."""
  import sys

  from test_oracle import _bits_to_f32, _triton_dropout_cases, dropout_fixture_limit

  sys.path.insert(0, GOLDEN)
  n = 0
  for case, (q, k, v, bias), o_ref_bits, lse_ref in _triton_dropout_cases():
    name, D, causal, dtype, p, seed, offset = case[0], case[6], case[7], case[9], case[11], case[12], case[13]
    tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
    qt, kt, vt = (torch.from_numpy(a.view(np.int16).copy()).view(tdt).cuda() for a in (q, k, v))
    bt = None if bias is None else torch.from_numpy(bias.view(np.int16).copy()).view(tdt).cuda()
    plan = {}
    o, lse = hip.forward(qt, kt, vt, bt, causal, D ** -0.5, dropout_p=p, philox_seed=seed, philox_offset=offset, plan_out=plan)
    assert "DROP=1" in plan["kernel"], plan
    want = _bits_to_f32(o_ref_bits, dtype)
    d = np.abs(_f32(o) - want)
    lim = dropout_fixture_limit(want, dtype, p)
    assert bool((d <= lim).all()) and d.mean() <= (6e-5 if dtype == "fp16" else 5e-4) / (1.0 - p), (name, d.max(), d.mean())
    assert (lse.cpu() - torch.from_numpy(lse_ref)).abs().max().item() <= 3e-5, name
    o_shift, _ = hip.forward(qt, kt, vt, bt, causal, D ** -0.5, dropout_p=p, philox_seed=seed, philox_offset=offset + 1)
    assert (np.abs(_f32(o_shift) - want) > lim).mean() > 0.5, name  # the check can tell another stream
    n += 1
  assert n == 3


def test_short_query_kernels_match_the_executed_reference_split_kv_decode_path(hip):
  """The reference's split-KV decode kernels executed with 3 / 4 / 5 splits (tests/golden/ref_triton_decode.npz) vs the short-query launches of the HIP
  library (D split over the waves, KV split over workgroups + ffpa_fwd_merge_kernel — whatever count ITS plan takes, and with the reference's count forced)."""
  import sys

  from test_oracle import _bits_to_f32, _triton_decode_cases, decode_fixture_limits

  sys.path.insert(0, GOLDEN)
  n = 0
  for case, (q, k, v, bias), o_ref_bits, lse_ref in _triton_decode_cases():
    name, D, causal, dtype, splits = case[0], case[6], case[7], case[9], case[11]
    tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
    qt, kt, vt = (torch.from_numpy(a.view(np.int16).copy()).view(tdt).cuda() for a in (q, k, v))
    bt = None if bias is None else torch.from_numpy(bias.view(np.int16).copy()).view(tdt).cuda()
    want = _bits_to_f32(o_ref_bits, dtype)
    lim, mean_lim, lse_lim = decode_fixture_limits(want, case)
    for req in (0, splits):
      plan = {}
      o, lse = hip.forward(qt, kt, vt, bt, causal, D ** -0.5, num_splits=req, plan_out=plan)
      assert plan["variant"] == 1 and (req == 0 or 1 < plan["splits"] <= splits), plan
      d = np.abs(_f32(o) - want)
      assert bool((d <= lim).all()) and d.mean() <= mean_lim * 1.25, (name, req, plan, d.max(), d.mean())
      assert (lse.cpu() - torch.from_numpy(lse_ref)).abs().max().item() <= lse_lim, (name, req)
    n += 1
  assert n == 4


def test_public_api_matches_the_executed_reference_api_path(hip):
  """`ffpa_attn_func(q, k, v, attn_mask=..., is_causal=..., scale=..., enable_gqa=...)` of this package on the GPU vs the reference's own argument handling +
  Triton forward executed on the same user-level arguments (tests/golden/ref_triton_api.npz): 2-D / 3-D boolean masks, an fp32 mask broadcasting over rows,
  default and custom scale, GQA, tail-aligned causal — all at shapes the reference serves with its own kernels (no SDPA fallback on either side)."""
  import sys

  from ffpa_attn_amd import ffpa_attn_func
  from ffpa_attn_amd.functional import FFPAAttnMeta
  from test_oracle import _bits_to_f32, _triton_api_cases, triton_fixture_limits

  sys.path.insert(0, GOLDEN)
  n = 0
  for case, (q, k, v, mask_np), o_ref_bits, lse_ref, _ in _triton_api_cases():
    name, Nkv, causal, dtype, scale, gqa = case[0], case[5], case[7], case[9], case[10], case[11]
    tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
    qt, kt, vt = (torch.from_numpy(a.view(np.int16).copy()).view(tdt).cuda() for a in (q, k, v))
    mask = None if mask_np is None else torch.from_numpy(mask_np).cuda()
    assert not FFPAAttnMeta.from_kwargs().fallback(qt, kt, mask, 0.0), name  # the HIP kernel serves it, not SDPA
    o = ffpa_attn_func(qt, kt, vt, attn_mask=mask, is_causal=causal, scale=scale, enable_gqa=gqa)
    want = _bits_to_f32(o_ref_bits, dtype)
    lim, mean_lim = triton_fixture_limits(want, dtype, Nkv, None, None)
    d = np.abs(_f32(o) - want)
    assert bool((d <= lim).all()) and d.mean() <= mean_lim * (1.25 if dtype == "fp16" else 1.0), (name, d.max(), d.mean())
    n += 1
  assert n == 4


# ----------------------------------------------------------------------------- fast path == safe path
@pytest.mark.parametrize("D", [64])
def test_dma_and_transpose_read_path_is_bit_identical_to_register_staged_twin(hip, D):
  """Same arithmetic, two data paths: LDS-DMA + ds_read_b64_tr_b16 vs plain loads + scalar gathers (the 32x32x16 kernel, which serves
  the prefill launches of this head dim; the twins of the larger head dims are the independent second mapping the 16x16x32 kernel is
  checked against up to rounding: tests/test_m16_gpu.py)."""
  q, k, v = _rand((2, 4, 200, D), seed=1), _rand((2, 2, 333, D), seed=2), _rand((2, 2, 333, D), seed=3)
  for causal in (False, True):
    a, la = hip.forward(q, k, v, None, causal, 0.05)
    b, lb = hip.forward(q, k, v, None, causal, 0.05, flags=hip.FLAG_DEBUG_SAFE_PATH)
    assert torch.equal(a, b) and torch.equal(la, lb), (D, causal)


def test_xcd_remap_does_not_change_results(hip):
  q, k, v = _rand((2, 8, 700, 512), seed=4), _rand((2, 8, 900, 512), seed=5), _rand((2, 8, 900, 512), seed=6)
  a, _ = hip.forward(q, k, v, None, True, 0.044)
  b, _ = hip.forward(q, k, v, None, True, 0.044, flags=hip.FLAG_NO_XCD_REMAP)
  assert torch.equal(a, b)


# ----------------------------------------------------------------------------- shapes
@pytest.mark.parametrize("D", list(range(64, 1025, 64)) + [264, 1000])
def test_every_head_dim(hip, D):
  q, k, v = _rand((1, 2, 130, D), seed=D), _rand((1, 1, 257, D), seed=D + 1), _rand((1, 1, 257, D), seed=D + 2)
  o, lse = hip.forward(q, k, v, None, False, 1.0 / D ** 0.5)
  assert o.shape == q.shape
  bc = hip.tile_config(hip.padded_head_dim(D))["block_keys"]
  _check_vs_oracle(o, lse, q, k, v, block_keys=bc, name=f"D{D}")


@pytest.mark.parametrize("D", [8, 72, 200, 264, 328, 456, 504, 520, 648, 968, 1000, 1016, 1001])
def test_head_dims_between_the_built_multiples_of_64_run_in_kernel(hip, D):
  """A head dim that is a multiple of 8 runs on the next 64-multiple instantiation with its missing columns read as zeros
  (Q guard, K range-check zero-fill) and never stored — no padded copies of q / k / v.  Adding exact zeros to the QK^T sums
  and leaving O's extra columns out cannot change a bit: the result must EQUAL the host-padded run (the reference's way,
  csrc/cuffpa/ffpa_api.cc:123-161), for the prefill tiles, causal, a ragged last tile, the short-query split-KV path and
  dropout (per-piece DMA form).  The tensors are allocated exactly (the last row ends the allocation).  D = 1001: not a
  multiple of 8 — padded to 1008 by the host, then the same path."""
  Dk = hip.padded_head_dim(D)
  for (B, Hq, Hkv, Nq, Nkv, causal, drop) in ((1, 2, 1, 130, 257, False, 0.0), (1, 2, 2, 200, 333, True, 0.0), (2, 4, 2, 3, 1500, False, 0.0),
                                              (1, 2, 2, 96, 160, False, 0.25)):
    q, k, v = _rand((B, Hq, Nq, D), seed=D), _rand((B, Hkv, Nkv, D), seed=D + 1), _rand((B, Hkv, Nkv, D), seed=D + 2)
    kw = dict(dropout_p=drop, philox_seed=1234, philox_offset=8) if drop else {}
    o, lse = hip.forward(q, k, v, None, causal, D ** -0.5, **kw)
    assert o.shape == q.shape and o.is_contiguous()
    qp, kp, vp = (F.pad(t, (0, Dk - D)) for t in (q, k, v))
    op, lsep = hip.forward(qp, kp, vp, None, causal, D ** -0.5, **kw)
    assert torch.equal(o, op[..., :D]) and torch.equal(lse, lsep), (D, Nq, Nkv, causal, drop)
    if not drop:
      _check_vs_oracle(o, lse, q, k, v, causal=causal, block_keys=hip.tile_config(Dk)["block_keys"], name=f"D{D} {Nq}x{Nkv}")
  if D >= 264:
    from ffpa_attn_amd import ffpa_attn_func

    q, k, v = _rand((1, 4, 640, D), seed=3), _rand((1, 4, 640, D), seed=4), _rand((1, 4, 640, D), seed=5)
    _close(ffpa_attn_func(q, k, v), F.scaled_dot_product_attention(q, k, v), q.dtype, f"api D{D}")


@pytest.mark.parametrize("Nq,Nkv", [(1, 1), (1, 4096), (7, 513), (15, 64), (127, 129), (128, 128), (129, 65), (513, 1000),
                                   (1000, 31), (640, 2049), (64, 63), (65, 32), (33, 33)])
@pytest.mark.parametrize("D", [320, 512, 768])
def test_boundary_sequence_lengths(hip, Nq, Nkv, D):
  q, k, v = _rand((1, 2, Nq, D), seed=Nq), _rand((1, 2, Nkv, D), seed=Nkv + 7), _rand((1, 2, Nkv, D), seed=Nkv + 9)
  o, lse = hip.forward(q, k, v, None, False, 1.0 / D ** 0.5)
  bc = hip.tile_config(D)["block_keys"]
  _check_vs_oracle(o, lse, q, k, v, block_keys=bc, name=f"{Nq}x{Nkv}xD{D}")


@pytest.mark.parametrize("Hq,Hkv", [(8, 8), (8, 2), (8, 1), (6, 3)])
def test_gqa_mqa_vs_repeat_interleave(hip, Hq, Hkv):
  q, k, v = _rand((2, Hq, 300, 512), seed=11), _rand((2, Hkv, 420, 512), seed=12), _rand((2, Hkv, 420, 512), seed=13)
  o, _ = hip.forward(q, k, v, None, False, 0.0442)
  g = Hq // Hkv
  ref, _ = hip.forward(q, k.repeat_interleave(g, 1), v.repeat_interleave(g, 1), None, False, 0.0442)
  assert torch.equal(o, ref)  # same arithmetic, only the head mapping differs
  _check_vs_oracle(o, None, q, k, v, name=f"gqa{Hq}/{Hkv}")


def test_strided_views_are_honoured_without_copies(hip):
  B, H, N, D = 2, 4, 260, 512
  qs = _rand((B, N, H, D), seed=21)  # [B, N, H, D] storage, transposed view: the typical caller
  ks, vs = _rand((B, N, H, D), seed=22), _rand((B, N, H, D), seed=23)
  q, k, v = qs.transpose(1, 2), ks.transpose(1, 2), vs.transpose(1, 2)
  assert not q.is_contiguous()
  o, lse = hip.forward(q, k, v, None, True, 0.0442)
  oc, lc = hip.forward(q.contiguous(), k.contiguous(), v.contiguous(), None, True, 0.0442)
  assert torch.equal(o, oc) and torch.equal(lse, lc)
  # batch-expanded K/V (stride 0 over batch) — one KV cache shared by every batch element
  ke, ve = k[:1].expand(B, H, N, D), v[:1].expand(B, H, N, D)
  o2, _ = hip.forward(q, ke, ve, None, False, 0.0442)
  o3, _ = hip.forward(q, ke.contiguous(), ve.contiguous(), None, False, 0.0442)
  assert torch.equal(o2, o3)


# ----------------------------------------------------------------------------- causal
@pytest.mark.parametrize("Nq,Nkv", [(512, 512), (191, 191), (100, 1000), (513, 640), (1, 777), (129, 129)])
@pytest.mark.parametrize("D", [320, 512, 1024])
def test_causal_tail_aligned(hip, Nq, Nkv, D):
  q, k, v = _rand((1, 2, Nq, D), seed=31), _rand((1, 2, Nkv, D), seed=32), _rand((1, 2, Nkv, D), seed=33)
  o, lse = hip.forward(q, k, v, None, True, 1.0 / D ** 0.5)
  bc = hip.tile_config(D)["block_keys"]
  _check_vs_oracle(o, lse, q, k, v, causal=True, block_keys=bc, name=f"causal {Nq}x{Nkv} D{D}")
  rows = torch.arange(Nq, device="cuda")[:, None]
  cols = torch.arange(Nkv, device="cuda")[None, :]
  ref = F.scaled_dot_product_attention(q, k, v, attn_mask=cols <= rows + (Nkv - Nq))  # tests/test_ffpa_fwd.py:1268-1304
  _close(o, ref, q.dtype)


def test_causal_top_left_offset_matches_sdpa_is_causal(hip):
  """causal_offset = 0 is PyTorch's alignment; Nq > Nkv is BASELINE config 4's shape class."""
  q, k, v = _rand((1, 4, 700, 320), seed=41), _rand((1, 2, 300, 320), seed=42), _rand((1, 2, 300, 320), seed=43)
  o, lse = hip.forward(q, k, v, None, True, 320 ** -0.5, causal_offset=0)
  ref = F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True)
  _close(o, ref, q.dtype)
  _check_vs_oracle(o, lse, q, k, v, causal=True, causal_offset=0, name="topleft")


def test_negative_causal_offset_gives_nan_rows_like_a_fully_masked_sdpa_row(hip):
  q, k, v = _rand((1, 1, 200, 320), seed=44), _rand((1, 1, 150, 320), seed=45), _rand((1, 1, 150, 320), seed=46)
  o, lse = hip.forward(q, k, v, None, True, 320 ** -0.5)  # tail aligned with Nq > Nkv: rows 0..49 see nothing
  assert torch.isnan(o[0, 0, :50]).all() and torch.isinf(lse[0, 0, :50]).all()
  assert torch.isfinite(o[0, 0, 50:]).all()
  _check_vs_oracle(o, lse, q, k, v, causal=True, name="neg-offset")


# ----------------------------------------------------------------------------- masks / bias
@pytest.mark.parametrize("shape", [(1, 1, 1, 600), (2, 1, 1, 600), (1, 4, 1, 600), (1, 1, 520, 600), (2, 4, 520, 600),
                                   (2, 1, 520, 1), (1, 4, 520, 600)])
@pytest.mark.parametrize("kind", ["bool", "add_q", "add_f32"])
def test_masks_all_broadcast_shapes(hip, shape, kind):
  B, H, Nq, Nkv, D = 2, 4, 520, 600, 320
  q, k, v = _rand((B, H, Nq, D), seed=51), _rand((B, H, Nkv, D), seed=52), _rand((B, H, Nkv, D), seed=53)
  g = torch.Generator(device="cuda").manual_seed(sum(shape) * 7 + len(kind))
  if kind == "bool":
    mask = torch.rand(shape, device="cuda", generator=g) > 0.25
    mask[..., 0] = True
    bias = torch.zeros(shape, dtype=q.dtype, device="cuda").masked_fill(~mask, float("-inf"))
    sdpa_mask = mask
  elif kind == "add_q":
    bias = (torch.randn(shape, device="cuda", generator=g) * 0.5).to(q.dtype)
    sdpa_mask = bias
  else:
    bias = torch.randn(shape, device="cuda", generator=g) * 0.5
    sdpa_mask = bias
  o, lse = hip.forward(q, k, v, bias, False, D ** -0.5)
  # reference = explicit fp32 math: PyTorch-ROCm's fused SDPA mishandles some broadcast bf16 masks (a
  # [B,1,Nq,1] additive mask is a per-row constant, i.e. a softmax no-op, yet its result moves by 0.2)
  s_ref = (q.float() @ k.float().transpose(-1, -2)) * D ** -0.5 + bias.float()
  ref = (torch.softmax(s_ref, -1) @ v.float()).to(q.dtype)
  _close(o, ref, q.dtype, f"{shape} {kind}")
  if shape[-1] != 1:  # (a key-broadcast [B,1,Nq,1] mask: SDPA-ROCm's result is unreliable — off by 0.2 for additive bf16, and once
    #                    flaky for the boolean form in a full-suite run — fp32 math above and the oracle below are the references)
    _close(o, F.scaled_dot_product_attention(q, k, v, attn_mask=sdpa_mask), q.dtype, f"sdpa {shape} {kind}")
  _check_vs_oracle(o, lse, q, k, v, bias=_f32(bias), name=f"{shape} {kind}")


def test_partially_and_fully_masked_rows(hip):
  B, H, Nq, Nkv, D = 1, 2, 513, 300, 512
  q, k, v = _rand((B, H, Nq, D), seed=61), _rand((B, H, Nkv, D), seed=62), _rand((B, H, Nkv, D), seed=63)
  mask = torch.ones(1, 1, Nq, Nkv, dtype=torch.bool, device="cuda")
  mask[0, 0, 5, :] = False           # fully masked -> NaN row (SDPA semantics)
  mask[0, 0, 9, :128] = False        # masked for the first tiles only: must stay finite
  mask[0, 0, 300, 1:] = False        # a single visible key: O = V[0]
  bias = torch.zeros(mask.shape, dtype=q.dtype, device="cuda").masked_fill(~mask, float("-inf"))
  o, lse = hip.forward(q, k, v, bias, False, D ** -0.5)
  assert torch.isnan(o[:, :, 5]).all() and torch.isfinite(o[:, :, 9]).all()
  assert torch.equal(o[0, :, 300], v[0, :, 0])
  _check_vs_oracle(o, lse, q, k, v, bias=_f32(bias), name="masked rows")


# ----------------------------------------------------------------------------- lazy rescale
@pytest.mark.parametrize("D", [512, 1024])
def test_forced_rescale_branch_and_threshold_sweep(hip, D):
  """Spike keys far above everything before them, late in the sequence (cdna guide rule 26)."""
  q, k, v = _rand((1, 2, 256, D), seed=71), _rand((1, 2, 1500, D), seed=72), _rand((1, 2, 1500, D), seed=73)
  k = k.clone()
  for row, key, gain in ((3, 700, 0.6), (3, 1300, 1.3), (40, 1499, 0.9), (200, 65, 0.8)):
    k[0, :, key] = (q[0, :, row].float() * gain).to(k.dtype)
  scale = D ** -0.5
  bc = hip.tile_config(D)["block_keys"]
  outs = []
  for thr in (0.0, 8.0, 40.0):
    o, lse = hip.forward(q, k, v, None, False, scale, rescale_threshold=thr)
    _check_vs_oracle(o, lse, q, k, v, threshold=thr, block_keys=bc, name=f"thr{thr}")
    outs.append(o.float())
  assert (outs[0] - outs[1]).abs().max() < 1.6e-2 and (outs[0] - outs[2]).abs().max() < 1.6e-2
  s = (q[0, 0, 3].float() @ k[0, 0].float().T) * scale * 1.4427
  assert (s.max() - s[:64].max()).item() > 16  # the branch really fires


# ----------------------------------------------------------------------------- fp16
def test_fp16(hip):
  q, k, v = (_rand((1, 4, 600, 512), torch.float16, seed=s) for s in (81, 82, 83))
  o, lse = hip.forward(q, k, v, None, True, 512 ** -0.5)
  _check_vs_oracle(o, lse, q, k, v, causal=True, name="fp16")
  ref = F.scaled_dot_product_attention(q, k, v, is_causal=True)
  _close(o, ref, q.dtype)


# ----------------------------------------------------------------------------- BASELINE sizes
def _baseline_inputs(B, Hq, Hkv, Nq, Nkv, D):
  torch.manual_seed(0)  # "seed 0; q then k then v" (tests/test_ffpa_fwd.py:116-121)
  q = torch.randn(B, Hq, Nq, D, dtype=torch.bfloat16, device="cuda")
  k = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  v = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  return q, k, v


@pytest.mark.parametrize("D", [512, 1024])
def test_baseline_config_2_and_3_full_size(hip, D):
  """B=1 H=32 N=8192 D=512 / 1024: vs SDPA on the same inputs (max-abs <= 1e-2), vs the oracle on
  sampled rows of sampled heads, determinism, and exact V-scaling."""
  q, k, v = _baseline_inputs(1, 32, 32, 8192, 8192, D)
  scale = D ** -0.5
  o, lse = hip.forward(q, k, v, None, False, scale)
  ref = F.scaled_dot_product_attention(q, k, v)
  err = (o.float() - ref.float()).abs()
  assert err.max().item() <= NORTH_STAR_MAX_ABS, err.max().item()
  _close(o, ref, q.dtype)
  bc = hip.tile_config(D)["block_keys"]
  # 8 heads x 64 rows = 512 rows against the oracle, the per-element allowance scaled by each row's largest p/l (~ 1e-4 here)
  for i, head in enumerate((0, 5, 9, 14, 17, 22, 27, 31)):
    r0 = (0, 1000, 2040, 3333, 4090, 5555, 7000, 8128)[i]
    sl = slice(head, head + 1)
    _check_vs_oracle(o[:, sl], lse[:, sl], q[:, sl], k[:, sl], v[:, sl], rows=(r0, r0 + 64), block_keys=bc, name=f"h{head}")
  o2, lse2 = hip.forward(q, k, v, None, False, scale)
  assert torch.equal(o, o2) and torch.equal(lse, lse2)          # run-to-run deterministic
  o4, _ = hip.forward(q, k, v * 4, None, False, scale)           # P unchanged, V scaled by 2^2: exact
  assert torch.equal(o4, o * 4)


def test_oracle_check_catches_a_dropped_kv_tile(hip):
  """The oracle comparison must bite at the BASELINE key count: an output computed WITHOUT the last KV tile (32 or 64 of 8192 keys: what a
  wrong tile bound would produce) has to fail it — per element AND in the mean — while the true output passes (round-3 review: with the
  constant 2.5e-3 allowance the element bound could not see this)."""
  for D in (512, 1024):
    q, k, v = _baseline_inputs(1, 2, 2, 256, 8192, D)
    bc = hip.tile_config(D)["block_keys"]
    scale = D ** -0.5
    o, lse = hip.forward(q, k, v, None, False, scale, num_splits=1)
    _check_vs_oracle(o[:, :1], lse[:, :1], q[:, :1], k[:, :1], v[:, :1], rows=(64, 128), block_keys=bc, name="intact")
    o_bad, lse_bad = hip.forward(q, k[:, :, : 8192 - bc].contiguous(), v[:, :, : 8192 - bc].contiguous(), None, False, scale, num_splits=1)
    with pytest.raises(AssertionError, match="max err"):
      _check_vs_oracle(o_bad[:, :1], None, q[:, :1], k[:, :1], v[:, :1], rows=(64, 128), block_keys=bc, name="dropped tile")


def test_baseline_config_4_gqa_cross_causal_mask(hip):
  """B=2 Hq=32/Hkv=8 Nq=8192 Nkv=2048 D=320, SDPA-style (top-left) causal mask — both as the
  structured causal_offset=0 path and as an explicit boolean mask through the bias path."""
  q, k, v = _baseline_inputs(2, 32, 8, 8192, 2048, 320)
  scale = 320 ** -0.5
  ref = F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True)
  o, lse = hip.forward(q, k, v, None, True, scale, causal_offset=0)
  _within_north_star(o, ref)
  late = slice(64, None)  # rows that average >= 64 keys: plain max-abs <= 1e-2
  assert (o[:, :, late].float() - ref[:, :, late].float()).abs().max().item() <= NORTH_STAR_MAX_ABS
  mask = torch.ones(8192, 2048, dtype=torch.bool, device="cuda").tril()
  bias = torch.zeros(1, 1, 8192, 2048, dtype=q.dtype, device="cuda").masked_fill(~mask, float("-inf"))
  om, _ = hip.forward(q, k, v, bias, False, scale)
  _within_north_star(om, ref)
  assert (om.float() - o.float()).abs().max().item() <= 8e-3  # (the 16-bit mask runs the 64-key bias-tile build: one bf16 ulp at |O| in [1, 2))
  bc4 = hip.tile_config(320)["block_keys"]
  for i, head in enumerate((0, 3, 8, 13, 18, 23, 28, 31)):  # 8 heads x 64 rows, both batch elements
    r0 = (0, 500, 1100, 2040, 3000, 4500, 6000, 8128)[i]
    sl, kv, bb = slice(head, head + 1), slice(head // 4, head // 4 + 1), slice(i % 2, i % 2 + 1)
    _check_vs_oracle(o[bb, sl], lse[bb, sl], q[bb, sl], k[bb, kv], v[bb, kv], causal=True, causal_offset=0, rows=(r0, r0 + 64),
                     block_keys=bc4, name=f"cfg4 h{head}")


def test_baseline_config_5_unsharded_on_one_gpu(hip):
  """B=8 H=32 N=8192 D=512 in ONE launch: 2^30 elements (2 GiB) per tensor, so batch / head offsets pass
  2^31 bytes.  Every (batch, head) unit must equal the same unit run alone (units are independent: this is
  what sharding over ranks relies on), also through a token-major [B,N,H,D] view (rows 32 KiB apart)."""
  B, H, N, D = 8, 32, 8192, 512
  g = torch.Generator(device="cuda").manual_seed(5)
  x = torch.randn(3, B, N, H, D, device="cuda", dtype=torch.bfloat16, generator=g)  # token-major storage
  q, k, v = (t.transpose(1, 2) for t in x)          # [B,H,N,D] views, row stride H*D
  scale = D ** -0.5
  o, lse = hip.forward(q, k, v, None, False, scale)
  assert o.shape == (B, H, N, D)
  for b, h in ((0, 0), (3, 17), (7, 31)):
    qs, ks, vs = (t[b:b + 1, h:h + 1].contiguous() for t in (q, k, v))
    o1, lse1 = hip.forward(qs, ks, vs, None, False, scale, num_splits=1)  # (a lone unit would otherwise get a KV-split plan)
    assert torch.equal(o[b:b + 1, h:h + 1], o1) and torch.equal(lse[b:b + 1, h:h + 1], lse1), (b, h)
  ref = F.scaled_dot_product_attention(q[7:8, 30:32], k[7:8, 30:32], v[7:8, 30:32])
  assert (o[7:8, 30:32].float() - ref.float()).abs().max().item() <= NORTH_STAR_MAX_ABS
  del x, q, k, v, o
  torch.cuda.empty_cache()


def test_kv_slices_larger_than_4_gib(hip):
  """Token-major [B,N,H,D] K/V with H*D = 16384 and N = 140k: each head's slice spans 4.6 GB, past any 32-bit
  offset from the slice base.  Tile-relative addressing must give exactly what dense per-head copies give."""
  N, H, D = 140_000, 32, 512
  g = torch.Generator(device="cuda").manual_seed(6)
  kv = torch.randn(2, 1, N, H, D, device="cuda", dtype=torch.bfloat16, generator=g)   # 2 x 4.6 GB
  k, v = (t.transpose(1, 2)[:, ::31] for t in kv)       # heads 0 and 31: [1,2,N,D] views, row stride 16384
  assert (k.size(2) - 1) * k.stride(2) * 2 >= 1 << 32
  q = torch.randn(1, 2, 256, D, device="cuda", dtype=torch.bfloat16, generator=g)
  scale = D ** -0.5
  o, lse = hip.forward(q, k, v, None, False, scale)
  od, lsed = hip.forward(q, k.contiguous(), v.contiguous(), None, False, scale)
  assert torch.equal(o, od) and torch.equal(lse, lsed)
  oc, _ = hip.forward(q, k, v, None, True, scale, causal_offset=N - 256)                # tail tiles at the far end
  ocd, _ = hip.forward(q, k.contiguous(), v.contiguous(), None, True, scale, causal_offset=N - 256)
  assert torch.equal(oc, ocd)
  q1 = q[:, :, :1]
  o1, _ = hip.forward(q1, k, v, None, False, scale)                                     # split-KV decode path
  o1d, _ = hip.forward(q1, k.contiguous(), v.contiguous(), None, False, scale)
  assert torch.equal(o1, o1d)
  ref = F.scaled_dot_product_attention(q, k.contiguous(), v.contiguous())
  assert (o.float() - ref.float()).abs().max().item() <= NORTH_STAR_MAX_ABS
  del kv, k, v
  torch.cuda.empty_cache()


def test_key_permutation_invariance_at_full_length(hip):
  q, k, v = _baseline_inputs(1, 2, 2, 1024, 8192, 512)
  o, lse = hip.forward(q, k, v, None, False, 512 ** -0.5)
  perm = torch.randperm(8192, device="cuda")
  op, lsep = hip.forward(q, k[:, :, perm], v[:, :, perm], None, False, 512 ** -0.5)
  assert (o.float() - op.float()).abs().max().item() <= 4e-3   # only summation order / rounding moves
  assert (lse - lsep).abs().max().item() <= 1e-4
  ones = torch.ones_like(v)
  oc, _ = hip.forward(q, k, ones, None, False, 512 ** -0.5)    # softmax rows sum to one
  assert (oc.float() - 1).abs().max().item() <= 2 ** -7


@pytest.mark.parametrize("D", [320, 512, 1024])
def test_a_constant_bias_moves_the_lse_and_nothing_else_at_full_length(hip, D):
  """softmax(s + c) = softmax(s): a constant additive bias — over the keys ([1,1,1,Nkv], the cached key-bias build), over rows and keys ([1,1,Nq,Nkv], the
  staged bias-tile build) — must leave O where the unmasked build puts it (to the rounding of P against a shifted running max) and move LSE by exactly c.
  Three builds of the prefill kernel against each other at the BASELINE key count."""
  q, k, v = _baseline_inputs(1, 2, 2, 1024, 8192, D)
  scale = D ** -0.5
  o, lse = hip.forward(q, k, v, None, False, scale)
  for c, shape in ((1.5, (1, 1, 1, 8192)), (-2.0, (1, 1, 1024, 8192))):
    bias = torch.full(shape, c, dtype=q.dtype, device="cuda")
    plan = {}
    ob, lseb = hip.forward(q, k, v, bias, False, scale, plan_out=plan)
    assert "MK=3" in plan["kernel"] or "MK=1" in plan["kernel"], plan
    d = (o.float() - ob.float()).abs()
    assert d.max().item() <= 5e-4 and d.mean().item() <= 3e-5, (D, shape, d.max().item(), d.mean().item())
    assert (lseb - lse - c).abs().max().item() <= 2e-4, (D, shape)


# ----------------------------------------------------------------------------- public API on the GPU
def test_public_api_routes_large_d_to_the_kernel(hip, monkeypatch):
  from ffpa_attn_amd import ffpa_attn_func

  q, k, v = _rand((1, 4, 640, 512), seed=91), _rand((1, 2, 900, 512), seed=92), _rand((1, 2, 900, 512), seed=93)
  calls = []
  real = torch._C._nn.scaled_dot_product_attention
  monkeypatch.setattr(torch._C._nn, "scaled_dot_product_attention", lambda *a, **kw: calls.append(1) or real(*a, **kw))
  out = ffpa_attn_func(q, k, v, is_causal=True, enable_gqa=True)
  assert not calls, "large-D must not reach native SDPA (tests/test_monkey_patch.py:127-133)"
  direct, _ = hip.forward(q, k, v, None, True, 512 ** -0.5)
  assert torch.equal(out, direct)
  mask = torch.rand(1, 1, 640, 900, device="cuda") > 0.3
  mask[..., 0] = True
  outm = ffpa_attn_func(q, k, v, attn_mask=mask, enable_gqa=True)
  monkeypatch.undo()
  _close(outm, F.scaled_dot_product_attention(q, k, v, attn_mask=mask, enable_gqa=True), q.dtype)
  # short / small-D shapes fall back (functional.py:717-724)
  small = _rand((1, 4, 640, 128), seed=94)
  assert torch.equal(ffpa_attn_func(small, small, small), F.scaled_dot_product_attention(small, small, small))
  short = _rand((1, 4, 100, 512), seed=95)
  assert torch.equal(ffpa_attn_func(short, k[:, :1].expand(1, 4, 900, 512), v[:, :1].expand(1, 4, 900, 512)),
                     F.scaled_dot_product_attention(short, k[:, :1].expand(1, 4, 900, 512), v[:, :1].expand(1, 4, 900, 512)))


def test_monkey_patched_sdpa_and_torch_compile(hip, monkeypatch):
  from ffpa_attn_amd import ffpa_attn_func

  q, k, v = (_rand((1, 4, 768, 320), seed=s) for s in (101, 102, 103))
  eager = ffpa_attn_func(q, k, v, is_causal=True)
  compiled = torch.compile(ffpa_attn_func, fullgraph=False)(q, k, v, is_causal=True)  # tests/test_ffpa_compile.py:52-89
  assert torch.equal(eager, compiled)
  monkeypatch.setattr(F, "scaled_dot_product_attention", ffpa_attn_func)
  assert torch.equal(F.scaled_dot_product_attention(q, k, v, is_causal=True), eager)


def test_op_schema_and_lse_contract(hip):
  q, k, v = (_rand((2, 4, 520, 320), seed=s) for s in (111, 112, 113))
  o, lse = torch.ops.ffpa_attn._fwd_hip(q, k, v, q.new_empty((0,)), 0, 1, 0, 320 ** -0.5, 0.0, 0, 0)
  assert o.shape == q.shape and o.dtype == q.dtype
  assert lse.shape == (2, 4, 520) and lse.dtype == torch.float32 and lse.is_contiguous()  # cuda/__init__.py:100-112
  s = (q.float() @ k.float().transpose(-1, -2)) * 320 ** -0.5
  assert (lse - torch.logsumexp(s, -1)).abs().max().item() < 2e-4


# ----------------------------------------------------------------------------- backward hookup (§8f rank 2)
@pytest.mark.parametrize("impl", ["auto", "recompute"])
@pytest.mark.parametrize("case", ["self", "gqa_causal_cross", "bias"])
def test_backward_through_saved_o_and_lse(hip, impl, case, monkeypatch):
  """Gradients from (O, LSE) of the HIP forward vs torch autograd through fp32 math attention
  (the reference checks its sdpa-backward the same way, tests/test_ffpa_bwd.py)."""
  from ffpa_attn_amd import backward as bw
  from ffpa_attn_amd import ffpa_attn_func

  D = 512
  if case == "self":
    Hq, Hkv, Nq, Nkv, causal = 4, 4, 640, 640, False
  elif case == "gqa_causal_cross":
    Hq, Hkv, Nq, Nkv, causal = 4, 2, 600, 777, True
  else:
    Hq, Hkv, Nq, Nkv, causal = 2, 2, 520, 640, False
  q = _rand((1, Hq, Nq, D), seed=121).requires_grad_()
  k = _rand((1, Hkv, Nkv, D), seed=122).requires_grad_()
  v = _rand((1, Hkv, Nkv, D), seed=123).requires_grad_()
  bias = None
  if case == "bias":
    bias = (_rand((1, 1, Nq, Nkv), seed=124) * 0.5).requires_grad_()
  go = _rand((1, Hq, Nq, D), seed=125)

  if impl == "recompute":
    orig = bw.attention_backward
    monkeypatch.setattr(bw, "attention_backward", lambda *a, **kw: orig(*a, **{**kw, "force": "recompute"}))
  out = ffpa_attn_func(q, k, v, attn_mask=bias, is_causal=causal, enable_gqa=(Hq != Hkv))
  grads = torch.autograd.grad(out, [q, k, v] + ([bias] if bias is not None else []), go)

  qf, kf, vf = (t.detach().float().requires_grad_() for t in (q, k, v))
  bf = bias.detach().float().requires_grad_() if bias is not None else None
  g = Hq // Hkv
  s = (qf @ kf.repeat_interleave(g, 1).transpose(-1, -2)) * D ** -0.5
  if bf is not None:
    s = s + bf
  if causal:
    rows = torch.arange(Nq, device="cuda")[:, None]
    cols = torch.arange(Nkv, device="cuda")[None, :]
    s = s.masked_fill(cols > rows + (Nkv - Nq), float("-inf"))
  ref = torch.softmax(s, -1) @ vf.repeat_interleave(g, 1)
  rgrads = torch.autograd.grad(ref, [qf, kf, vf] + ([bf] if bf is not None else []), go.float())
  for name, a, b in zip(("dq", "dk", "dv", "dbias"), grads, rgrads):
    scale = b.abs().max().item()
    err = (a.float() - b).abs().max().item()
    assert err <= 3e-2 * scale + 1e-3, f"{case}/{impl} {name}: err {err:.3e} vs max {scale:.3e}"


# ----------------------------------------------------------------------------- short-query / decode path (§8f rank 3)
@pytest.mark.parametrize("D", [128, 320, 512, 1024])
@pytest.mark.parametrize("Nq,Hq,Hkv", [(1, 8, 8), (1, 8, 2), (1, 32, 4), (2, 8, 2), (7, 4, 1), (7, 8, 8), (20, 4, 4), (32, 4, 2)])
def test_short_query_split_kv(hip, D, Nq, Hq, Hkv):
  """Nq <= 32: D split over all four waves, KV split over workgroups + LSE merge, GQA heads packed into
  rows for Nq <= 7 (the reference's decode tests: tests/test_ffpa_fwd.py:929-1018)."""
  Nkv = 3000
  q, k, v = _rand((2, Hq, Nq, D), seed=131), _rand((2, Hkv, Nkv, D), seed=132), _rand((2, Hkv, Nkv, D), seed=133)
  plan = {}
  o, lse = hip.forward(q, k, v, None, False, D ** -0.5, plan_out=plan)
  assert plan["variant"] == 1 and plan["splits"] > 1
  assert plan["packed"] == (Hq != Hkv and Nq <= 7 and (Hq // Hkv) * Nq <= 32)
  _check_vs_oracle(o, lse, q, k, v, block_keys=_short_query_keys(hip, D), name=f"short {Nq}x{Hq}/{Hkv} D{D}", split=True)
  o1, lse1 = hip.forward(q, k, v, None, False, D ** -0.5, num_splits=1)  # unsplit: same answer up to rounding
  assert (o.float() - o1.float()).abs().max().item() <= 4e-3
  assert (lse - lse1).abs().max().item() <= 1e-4
  o3, _ = hip.forward(q, k, v, None, False, D ** -0.5, num_splits=3, plan_out=plan)
  assert 1 < plan["splits"] <= 3
  assert (o3.float() - o1.float()).abs().max().item() <= 4e-3


@pytest.mark.parametrize("D", [128, 320, 512, 640, 1024])
def test_streaming_kv_fetch_changes_no_bit(hip, D):
  """Short-query launches fetch K / V with the non-temporal hint when every byte has one reader and K + V exceed the Infinity Cache (ffpa_capi.hip;
  profiles/r04_kv_stream.txt).  A cache hint moves no data differently: forced on, forced off and the launch side's own choice give the same bits —
  MHA, packed and un-packed GQA, tails, causal, every short-query tile family."""
  for (B, Hq, Hkv, Nq, Nkv, causal, dt) in [(2, 8, 8, 1, 3001, False, torch.bfloat16), (1, 8, 2, 1, 2048, False, torch.float16), (1, 8, 2, 16, 1500, True, torch.bfloat16),
                                             (1, 4, 4, 32, 777, False, torch.bfloat16)]:
    q, k, v = _rand((B, Hq, Nq, D), dt, seed=601), _rand((B, Hkv, Nkv, D), dt, seed=602), _rand((B, Hkv, Nkv, D), dt, seed=603)
    plan = {}
    o0, l0 = hip.forward(q, k, v, None, causal, D ** -0.5, plan_out=plan)
    assert plan["variant"] == 1
    o1, l1 = hip.forward(q, k, v, None, causal, D ** -0.5, flags=hip.FLAG_KV_STREAM)
    o2, l2 = hip.forward(q, k, v, None, causal, D ** -0.5, flags=hip.FLAG_NO_KV_STREAM)
    assert torch.equal(o1, o2) and torch.equal(l1, l2) and torch.equal(o0, o1) and torch.equal(l0, l1), (D, B, Hq, Hkv, Nq, Nkv)
  _check_vs_oracle(o1, l1, q, k, v, block_keys=_short_query_keys(hip, D), name=f"streamed D{D}", split=True)


def test_split_partials_merged_inside_the_launch_equal_the_merge_kernel(hip):
  """Short-query KV-split launches with ffpa_fwd_params.split_tickets: the last split of a row tile to arrive merges the partials itself
  (write-through partial stores, one relaxed agent-scope ticket, agent-scope acquire by the merger: one launch per call).  Same arithmetic
  as ffpa_fwd_merge_kernel: the same bits, call after call with changing inputs and the workspace recycled (a stale partial from the
  previous call, or a split missed by the merger, would show in some word), with ragged tails and with another kernel keeping the chip
  busy.  Opt-in (it measured slower than the merge kernel: profiles/r03_split_merge.txt); prefill-tile splits always take the merge kernel."""
  shapes = [(8, 32, 8, 1, 8192 + 77, 512, False), (1, 32, 32, 1, 8192, 512, False), (3, 8, 8, 20, 5000, 320, True), (2, 16, 4, 7, 4097, 1024, False),
            (1, 8, 8, 1, 3000, 128, False), (4, 8, 2, 32, 6000, 640, True)]
  busy_a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
  for (B, Hq, Hkv, Nq, Nkv, D, causal) in shapes:
    k, v = _rand((B, Hkv, Nkv, D), seed=302), _rand((B, Hkv, Nkv, D), seed=303)
    plan = {}
    for it in range(6):
      q = _rand((B, Hq, Nq, D), seed=310 + it)
      if it % 2:
        busy_a @ busy_a  # something else on the chip while the splits arrive
      o1, l1 = hip.forward(q, k, v, None, causal, D ** -0.5, merge_in_launch=True, plan_out=plan)
      assert plan["splits"] > 1 and plan["variant"] == 1 and "in-launch" in plan["kernel"], plan
      o2, l2 = hip.forward(q, k, v, None, causal, D ** -0.5, merge_in_launch=False, plan_out=plan)
      assert plan["kernel"].endswith("ffpa_fwd_merge_kernel"), plan
      assert torch.equal(o1, o2) and torch.equal(l1, l2), ((B, Hq, Hkv, Nq, Nkv, D, causal), it)
  # the tickets are back to zero after every launch: the buffer of this stream holds nothing but zeros
  assert hip._TICKETS
  for t in hip._TICKETS.values():
    assert int(t.abs().max()) == 0
  # prefill-tile splits (underfilled launches) ignore the tickets: one workgroup merging a 128-row tile would serialise the merge
  q, k, v = _rand((1, 4, 512, 512), seed=321), _rand((1, 4, 16384, 512), seed=322), _rand((1, 4, 16384, 512), seed=323)
  hip.forward(q, k, v, None, False, 512 ** -0.5, merge_in_launch=True, plan_out=plan)
  assert plan["variant"] == 0 and plan["splits"] > 1 and plan["kernel"].endswith("ffpa_fwd_merge_kernel"), plan


@pytest.mark.parametrize("Nq,Hq,Hkv", [(1, 8, 2), (7, 4, 1), (5, 8, 8), (24, 4, 4)])
def test_short_query_causal_and_tails(hip, Nq, Hq, Hkv):
  D, Nkv = 512, 1111
  q, k, v = _rand((1, Hq, Nq, D), seed=141), _rand((1, Hkv, Nkv, D), seed=142), _rand((1, Hkv, Nkv, D), seed=143)
  o, lse = hip.forward(q, k, v, None, True, D ** -0.5)                       # tail aligned
  _check_vs_oracle(o, lse, q, k, v, causal=True, block_keys=_short_query_keys(hip, q.size(-1)), name="short causal tail", split=True)
  o0, lse0 = hip.forward(q, k, v, None, True, D ** -0.5, causal_offset=0)    # top-left: row i sees keys 0..i
  _check_vs_oracle(o0, lse0, q, k, v, causal=True, causal_offset=0, block_keys=_short_query_keys(hip, q.size(-1)), name="short causal topleft", split=True)
  assert torch.equal(o0[:, :, 0], v[:, :, 0].repeat_interleave(Hq // Hkv, 1))  # row 0 sees only key 0


@pytest.mark.parametrize("D,Hq,Hkv,Nq,Nkv,causal", [(512, 4, 4, 512, 16384, False), (512, 8, 2, 640, 9000, True), (320, 2, 1, 1024, 8192, True),
                                                    (1024, 2, 2, 512, 8192, False)])
def test_underfilled_prefill_splits_the_kv_axis(hip, D, Hq, Hkv, Nq, Nkv, causal):
  """Prefill tiles with few workgroups (chunked prefill, long context, few heads): the plan splits the KV axis and
  merges by LSE.  Must agree with the unsplit launch to rounding and with SDPA at the north-star bound."""
  q, k, v = _rand((1, Hq, Nq, D), seed=401), _rand((1, Hkv, Nkv, D), seed=402), _rand((1, Hkv, Nkv, D), seed=403)
  scale = D ** -0.5
  plan = {}
  o, lse = hip.forward(q, k, v, None, causal, scale, plan_out=plan)
  assert plan["variant"] == 0 and plan["splits"] > 1, plan
  o1, lse1 = hip.forward(q, k, v, None, causal, scale, num_splits=1)
  assert (o.float() - o1.float()).abs().max().item() <= 4e-3 and (lse - lse1).abs().max().item() <= 1e-4
  g = Hq // Hkv
  mask = None
  if causal:
    rows, cols = torch.arange(Nq, device="cuda")[:, None], torch.arange(Nkv, device="cuda")[None, :]
    mask = cols <= rows + (Nkv - Nq)
  ref = F.scaled_dot_product_attention(q, k.repeat_interleave(g, 1), v.repeat_interleave(g, 1), attn_mask=mask)
  assert (o.float() - ref.float()).abs().max().item() <= NORTH_STAR_MAX_ABS
  bias = (_rand((1, 1, Nq, Nkv), seed=404) * 0.5)
  ob, _ = hip.forward(q, k, v, bias, False, scale, plan_out=plan)
  assert plan["splits"] > 1
  ob1, _ = hip.forward(q, k, v, bias, False, scale, num_splits=1)
  assert (ob.float() - ob1.float()).abs().max().item() <= 4e-3


@pytest.mark.parametrize("D,H,Nq,Nkv", [(512, 9, 4096, 8192), (512, 11, 4096, 8190), (1024, 5, 4096, 8192), (320, 10, 4096, 8192), (512, 5, 4096, 8192)])
def test_ragged_round_prefill_splits_the_kv_axis(hip, D, H, Nq, Nkv):
  """A launch of a little over one round of workgroups (1 < workgroups / CUs <= 1.5), or of part of one, on a long context: the plan splits the KV axis in 2
  or 3 (ffpa_capi.hip make_plan; profiles/r04_launch_side.txt).  Same answer as the unsplit launch to rounding, the oracle's bound on a
  row subset, and one split on request."""
  cus = torch.cuda.get_device_properties(0).multi_processor_count
  q, k, v = _rand((1, H, Nq, D), seed=411), _rand((1, H, Nkv, D), seed=412), _rand((1, H, Nkv, D), seed=413)
  scale = D ** -0.5
  plan = {}
  o, lse = hip.forward(q, k, v, None, False, scale, plan_out=plan)
  wgs = H * (Nq // plan["block_rows"])
  if not (0.5 * cus < wgs <= 1.5 * cus) or wgs == cus:
    pytest.skip(f"{wgs} workgroups on {cus} CUs is not a ragged round on this device")
  assert plan["variant"] == 0 and plan["splits"] in (2, 3), plan
  o1, lse1 = hip.forward(q, k, v, None, False, scale, num_splits=1, plan_out=plan)
  assert plan["splits"] == 1
  assert (o.float() - o1.float()).abs().max().item() <= 4e-3 and (lse - lse1).abs().max().item() <= 1e-4
  rows = torch.arange(0, Nq, Nq // 64, device="cuda")  # 64 rows spread over the row tiles, every head
  _check_vs_oracle(o[:, :, rows].contiguous(), lse[:, :, rows].contiguous(), q[:, :, rows].contiguous(), k, v, block_keys=plan["block_keys"],
                   name=f"ragged D{D} H{H}", split=True)


@pytest.mark.parametrize("D", [512, 320, 1024])
def test_mask_derived_tile_clipping_changes_nothing_but_the_time(hip, D):
  """kv_bounds (visible-key range per 32-row block, derived from the mask): tiles the mask hides entirely are
  skipped.  Results must be BIT-identical to walking every tile — for causal, sliding-window, padding and
  batch/head-dependent masks, rows without any visible key (NaN) included."""
  B, Hq, Hkv, Nq, Nkv = 2, 4, 2, 700, 1500
  q, k, v = _rand((B, Hq, Nq, D), seed=501), _rand((B, Hkv, Nkv, D), seed=502), _rand((B, Hkv, Nkv, D), seed=503)
  scale = D ** -0.5
  rows, cols = torch.arange(Nq, device="cuda")[:, None], torch.arange(Nkv, device="cuda")[None, :]
  masks = {
      "causal": cols <= rows + (Nkv - Nq),
      "window": (cols <= rows + 400) & (cols >= rows + 100),
      "padding": (cols < 900).expand(Nq, Nkv),
      "holes": ((cols <= rows + 300) & (rows >= 64)),                     # first 64 rows see nothing -> NaN rows
  }
  for name, m in masks.items():
    bias = torch.zeros(1, 1, Nq, Nkv, dtype=q.dtype, device="cuda").masked_fill(~m, float("-inf"))
    o_all, lse_all = hip.forward(q, k, v, bias, False, scale, kv_bounds=False)
    o_clip, lse_clip = hip.forward(q, k, v, bias, False, scale, kv_bounds=True)
    assert torch.equal(torch.nan_to_num(o_all.float(), nan=7.0), torch.nan_to_num(o_clip.float(), nan=7.0)), name
    assert torch.equal(lse_all, lse_clip) or torch.equal(torch.nan_to_num(lse_all, nan=7.0), torch.nan_to_num(lse_clip, nan=7.0)), name
    assert torch.isnan(o_all).any().item() == (name == "holes")
  # per-(batch, head) masks and precomputed bounds
  mbh = torch.rand(B, Hq, 1, Nkv, device="cuda") > 0.5
  mbh = (mbh & (cols[None, None] < 1000)).expand(B, Hq, Nq, Nkv).contiguous()
  mbh[..., 0] = True
  bias = torch.zeros(B, Hq, Nq, Nkv, dtype=q.dtype, device="cuda").masked_fill(~mbh, float("-inf"))
  bounds = hip.mask_kv_bounds(bias, Nq, Nkv)
  assert bounds.shape == (B, Hq, (Nq + 31) // 32, 4) and int(bounds[..., 1].max()) <= 1000
  o_all, _ = hip.forward(q, k, v, bias, False, scale, kv_bounds=False)
  o_clip, _ = hip.forward(q, k, v, bias, False, scale, kv_bounds=bounds)
  assert torch.equal(o_all, o_clip)
  ref = F.scaled_dot_product_attention(q, k, v, attn_mask=mbh, enable_gqa=True)
  _close(o_clip, ref, q.dtype)


def test_public_api_clips_masked_tiles_automatically(hip, monkeypatch):
  """ffpa_attn_func(attn_mask=<bool mask>) derives the bounds by itself (a [Nq, Nkv] mask shared by all heads); same
  bits as with the scan disabled, and the scan really ran."""
  from ffpa_attn_amd import ffpa_attn_func
  import ffpa_attn_amd.hip as hmod
  q, k, v = _rand((1, 8, 1024, 512), seed=601), _rand((1, 8, 2048, 512), seed=602), _rand((1, 8, 2048, 512), seed=603)
  rows, cols = torch.arange(1024, device="cuda")[:, None], torch.arange(2048, device="cuda")[None, :]
  mask = (cols <= rows + 1024) & (cols + 512 >= rows)
  calls = []
  real = hmod.mask_kv_bounds
  monkeypatch.setattr(hmod, "mask_kv_bounds", lambda *a, **kw: calls.append(1) or real(*a, **kw))
  out = ffpa_attn_func(q, k, v, attn_mask=mask)
  assert calls == [1]
  monkeypatch.setenv("FFPA_HIP_MASK_BOUNDS", "0")
  out_all = ffpa_attn_func(q, k, v, attn_mask=mask)
  assert calls == [1] and torch.equal(out, out_all)
  _close(out, F.scaled_dot_product_attention(q, k, v, attn_mask=mask), q.dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_mask_bounds_kernel_matches_the_torch_restatement(hip, dtype):
  """ffpa_attn_mask_kv_bounds (one fused pass on the GPU) vs the torch-op statement of the same thing on the CPU."""
  g = torch.Generator(device="cuda").manual_seed(7)
  for (bb, hb, nq, nkv, rows_bcast) in ((1, 1, 700, 1500, False), (2, 3, 97, 1111, False), (1, 2, 64, 5000, True), (1, 1, 33, 31, False)):
    shape = (bb, hb, 1 if rows_bcast else nq, nkv)
    keep = torch.rand(shape, device="cuda", generator=g) > 0.7
    r = torch.arange(shape[2], device="cuda")[:, None]
    c = torch.arange(nkv, device="cuda")[None, :]
    keep = keep & (c <= r * 2 + nkv // 3) & (c >= r // 2)               # banded + random holes
    if not rows_bcast and nq > 40:
      keep[..., 32:64, :] = False                                       # an empty 32-row block
    bias = (torch.randn(shape, device="cuda", generator=g) * 0.1).to(dtype).masked_fill(~keep, float("-inf"))
    got = hip.mask_kv_bounds(bias, nq, nkv)
    want = hip.mask_kv_bounds(bias.cpu(), nq, nkv)
    assert got.dtype == torch.int32 and torch.equal(got.cpu(), want), (dtype, shape)
  strided = torch.zeros(1, 1, 128, 2048, dtype=dtype, device="cuda").masked_fill(
      torch.arange(2048, device="cuda")[None, :] > torch.arange(128, device="cuda")[:, None] * 4 + 77, float("-inf"))
  view = strided[..., ::2]                                               # key stride 2
  assert torch.equal(hip.mask_kv_bounds(view, 128, 1024).cpu(), hip.mask_kv_bounds(view.cpu(), 128, 1024))


def test_short_query_bias_is_not_packed(hip):
  q, k, v = _rand((1, 8, 3, 512), seed=151), _rand((1, 2, 900, 512), seed=152), _rand((1, 2, 900, 512), seed=153)
  bias = (torch.randn(1, 8, 3, 900, device="cuda") * 0.5).to(q.dtype)
  plan = {}
  o, lse = hip.forward(q, k, v, bias, False, 512 ** -0.5, plan_out=plan)
  assert plan["variant"] == 1 and not plan["packed"]
  _check_vs_oracle(o, lse, q, k, v, bias=_f32(bias), block_keys=_short_query_keys(hip, q.size(-1)), name="short bias")


def test_decode_through_public_api(hip):
  from ffpa_attn_amd import ffpa_attn_func

  q, k, v = _rand((2, 16, 1, 512), seed=161), _rand((2, 4, 4096, 512), seed=162), _rand((2, 4, 4096, 512), seed=163)
  out = ffpa_attn_func(q, k, v, enable_gqa=True)
  ref = F.scaled_dot_product_attention(q, k, v, enable_gqa=True)
  _close(out, ref, q.dtype)


# ----------------------------------------------------------------------------- dropout (§8f rank 4)
@pytest.mark.parametrize("D,Nq,Nkv,Hq,Hkv,causal", [(512, 300, 777, 4, 2, False), (320, 129, 515, 2, 2, True),
                                                    (1024, 100, 300, 2, 1, False), (512, 5, 900, 8, 2, False)])
def test_dropout_mask_matches_the_oracle_bit_for_bit(hip, D, Nq, Nkv, Hq, Hkv, causal):
  """Same Philox seed/offset => same mask as the CPU restatement (prefill.cuh:398-546).  A single flipped
  mask bit would move O by ~p_k/l >> tolerance at these lengths."""
  q, k, v = _rand((2, Hq, Nq, D), seed=171), _rand((2, Hkv, Nkv, D), seed=172), _rand((2, Hkv, Nkv, D), seed=173)
  for (p, seed, off) in ((0.1, 1234567, 0), (0.5, 0xFEDCBA9876543210, 1_000_003)):  # offset % 4 != 0 too
    o, lse = hip.forward(q, k, v, None, causal, D ** -0.5, dropout_p=p, philox_seed=seed, philox_offset=off)
    qb, dt = fo.torch_to_bits(q)
    kb, _ = fo.torch_to_bits(k)
    vb, _ = fo.torch_to_bits(v)
    bc = _short_query_keys(hip, D) if Nq <= 32 else (32 if D > 512 else 64)
    _, o32, lse_ref = fo.oracle_forward(qb, kb, vb, dt, causal=causal, block_keys=bc, dropout_p=p, philox_seed=seed,
                                        philox_offset=off)
    err = np.abs(_f32(o) - o32)
    assert err.max() <= 2.0 ** -8 * np.abs(o32).max() + 6e-3 and err.mean() < 6e-4, (p, err.max(), err.mean())
    np.testing.assert_allclose(_f32(lse), lse_ref, atol=2e-4, rtol=2e-5)   # LSE is undropped
    o_nodrop, _ = hip.forward(q, k, v, None, causal, D ** -0.5)
    assert (o.float() - o_nodrop.float()).abs().max().item() > 0.02            # dropout really happened


@pytest.mark.parametrize("D,Nq", [(64, 96), (128, 96), (512, 1)])
@pytest.mark.parametrize("p", [2.0 ** -24, 1e-9, 0.25, 0.75, 0.999, float(np.nextafter(np.float32(0.5), np.float32(1.0)))])
def test_dropout_keep_threshold_is_the_float_comparison(hip, D, Nq, p):
  """The kernels compare the Philox word with an integer threshold the launch side derives from `((float)word + 1) * 2^-32 > p`
  (prefill.cuh:437-440; ffpa_capi.hip dropout_keep_threshold): at probabilities that sit on or next to a rounding boundary of that
  expression the mask is still the oracle's, in the 32x32x16 build (D = 64), the 16x16x32 build and the split-KV tiles (Nq = 1)."""
  Nkv = 320
  q, k, v = _rand((1, 2, Nq, D), seed=181), _rand((1, 2, Nkv, D), seed=182), _rand((1, 2, Nkv, D), seed=183)
  o, _ = hip.forward(q, k, v, None, False, D ** -0.5, dropout_p=p, philox_seed=0x1234567890ABCDEF, philox_offset=4)
  qb, dt = fo.torch_to_bits(q)
  kb, _ = fo.torch_to_bits(k)
  vb, _ = fo.torch_to_bits(v)
  _, o32, _ = fo.oracle_forward(qb, kb, vb, dt, causal=False, block_keys=_short_query_keys(hip, D) if Nq <= 32 else 64, dropout_p=p, philox_seed=0x1234567890ABCDEF, philox_offset=4)
  err = np.abs(_f32(o) - o32)
  # one flipped keep bit moves a row by about |v| / Nkv / (1 - p): far outside this bound at every p above
  assert err.max() <= (2.0 ** -7 * np.abs(o32).max() + 2e-3), (p, err.max(), np.abs(o32).max())


def test_dropout_public_api_reserves_generator_offsets(hip):
  from ffpa_attn_amd import ffpa_attn_func

  q, k, v = (_rand((1, 4, 600, 512), seed=s) for s in (181, 182, 183))
  torch.cuda.manual_seed(99)
  off0 = torch.cuda._get_rng_state_offset()
  a = ffpa_attn_func(q, k, v, dropout_p=0.2)
  assert torch.cuda._get_rng_state_offset() - off0 == (1 * 4 * 600 * 600 + 3) // 4 * 4   # functional.py:535-540
  b = ffpa_attn_func(q, k, v, dropout_p=0.2)
  assert not torch.equal(a, b)                 # generator advanced
  torch.cuda.manual_seed(99)
  assert torch.equal(ffpa_attn_func(q, k, v, dropout_p=0.2), a)   # same seed + offset => same mask
  ones = torch.ones_like(v)
  o = ffpa_attn_func(q * 0, k, ones, dropout_p=0.25)            # uniform attention: O = kept/N/(1-p)
  assert abs(o.float().mean().item() - 1.0) < 0.02


def test_dropout_backward_uses_the_forward_mask(hip):
  """dq/dk/dv under dropout vs autograd through explicit math with the SAME Philox mask."""
  from ffpa_attn_amd import ffpa_attn_func
  from ffpa_attn_amd.philox import dropout_keep_mask

  D, Hq, Hkv, Nq, Nkv, p = 512, 4, 2, 520, 640, 0.2
  q = _rand((1, Hq, Nq, D), seed=191).requires_grad_()
  k = _rand((1, Hkv, Nkv, D), seed=192).requires_grad_()
  v = _rand((1, Hkv, Nkv, D), seed=193).requires_grad_()
  go = _rand((1, Hq, Nq, D), seed=194)
  torch.cuda.manual_seed(5)
  seed, off = torch.cuda.initial_seed(), torch.cuda._get_rng_state_offset()
  out = ffpa_attn_func(q, k, v, dropout_p=p, enable_gqa=True)
  grads = torch.autograd.grad(out, [q, k, v], go)

  qf, kf, vf = (t.detach().float().requires_grad_() for t in (q, k, v))
  g = Hq // Hkv
  s = (qf @ kf.repeat_interleave(g, 1).transpose(-1, -2)) * D ** -0.5
  idx = ((torch.arange(Hq, device="cuda").view(1, Hq, 1, 1) * Nq + torch.arange(Nq, device="cuda").view(1, 1, Nq, 1)) * Nkv
         + torch.arange(Nkv, device="cuda").view(1, 1, 1, Nkv))
  keep = dropout_keep_mask(seed, off, idx, p).float() / (1 - p)
  ref = (torch.softmax(s, -1) * keep) @ vf.repeat_interleave(g, 1)
  assert (out.float() - ref).abs().max().item() < 2e-2
  rgrads = torch.autograd.grad(ref, [qf, kf, vf], go.float())
  for name, a, b in zip(("dq", "dk", "dv"), grads, rgrads):
    scale = b.abs().max().item()
    assert (a.float() - b).abs().max().item() <= 3e-2 * scale + 1e-3, name


# ----------------------------------------------------------------------------- randomized sweep
def _random_case(rng):
  D = int(rng.choice([64, 128, 192, 264, 320, 384, 448, 512, 520, 576, 640, 768, 1000, 1024]))
  Hkv = int(rng.choice([1, 2, 3]))
  group = int(rng.choice([1, 1, 2, 4]))
  Nq = int(rng.choice([1, 2, 7, 9, 31, 32, 33, 64, 100, 129, 257, 300]))
  Nkv = int(rng.choice([1, 17, 32, 33, 63, 64, 65, 127, 200, 333, 512, 700]))
  mode = rng.choice(["plain", "causal_tail", "causal_topleft", "causal_off", "bias_bool", "bias_add", "bias_f32", "dropout"])
  dtype = torch.float16 if rng.random() < 0.25 else torch.bfloat16
  strided = rng.random() < 0.3
  return dict(B=int(rng.choice([1, 2])), Hq=Hkv * group, Hkv=Hkv, Nq=Nq, Nkv=Nkv, D=D, mode=str(mode), dtype=dtype, strided=strided)


# FFPA_FUZZ_SEEDS=a:b widens the sweep for a one-off fuzz run (the default 48 cases keep the suite short); FFPA_FUZZ_FLAGS=<int> ORs launch flags into every
# call of the sweep (0x1000 = FFPA_FLAG_WIDE_TILE: the cases the wide-row tile has a build for — D in (256, 320], no additive bias, no dropout — run it
# whatever their size; the plan's own rule only takes it for thousands of workgroups)
_FUZZ = os.environ.get("FFPA_FUZZ_SEEDS", "0:48").split(":")
_FUZZ_FLAGS = int(os.environ.get("FFPA_FUZZ_FLAGS", "0"), 0)
# FFPA_FUZZ_SPLITS=<n> with FFPA_FUZZ_FLAGS=0x100040 (FFPA_FLAG_TILE_RANGES | FFPA_FLAG_FORCE_SPLITS): every causal case the dense mode of the packed-sequence
# kernel has a build for (no bias / dropout, offset >= 0, two row tiles or more) splits every row tile's own visible KV tiles into n ranges; the others run one range
_FUZZ_SPLITS = int(os.environ.get("FFPA_FUZZ_SPLITS", "0"), 0)


@pytest.mark.parametrize("seed", range(int(_FUZZ[0]), int(_FUZZ[1])))
def test_randomized_against_oracle(hip, seed):
  rng = np.random.default_rng(1000 + seed)
  c = _random_case(rng)
  B, Hq, Hkv, Nq, Nkv, D, dt = c["B"], c["Hq"], c["Hkv"], c["Nq"], c["Nkv"], c["D"], c["dtype"]
  if c["strided"]:  # [B, N, H, D] storage viewed as [B, H, N, D]
    q = _rand((B, Nq, Hq, D), dt, seed=seed * 3 + 1).transpose(1, 2)
    k = _rand((B, Nkv, Hkv, D), dt, seed=seed * 3 + 2).transpose(1, 2)
    v = _rand((B, Nkv, Hkv, D), dt, seed=seed * 3 + 3).transpose(1, 2)
  else:
    q, k, v = _rand((B, Hq, Nq, D), dt, seed=seed * 3 + 1), _rand((B, Hkv, Nkv, D), dt, seed=seed * 3 + 2), _rand((B, Hkv, Nkv, D), dt, seed=seed * 3 + 3)
  kw, okw, bias = {}, {}, None
  causal = c["mode"].startswith("causal")
  if c["mode"] == "causal_topleft":
    kw["causal_offset"] = okw["causal_offset"] = 0
  elif c["mode"] == "causal_off":
    off = int(rng.integers(-3, Nkv + 2))
    kw["causal_offset"] = okw["causal_offset"] = off
  elif c["mode"] == "bias_bool":
    m = torch.rand((B, 1, 1, Nkv), device="cuda") > 0.3
    m[..., 0] = True
    bias = torch.zeros(m.shape, dtype=dt, device="cuda").masked_fill(~m, float("-inf"))
  elif c["mode"] == "bias_add":
    bias = (torch.randn((1, Hq, Nq, Nkv), device="cuda") * 0.5).to(dt)
  elif c["mode"] == "bias_f32":
    bias = torch.randn((B, 1, Nq, 1), device="cuda") * 0.5 + torch.randn((1, 1, 1, Nkv), device="cuda")
  elif c["mode"] == "dropout":
    kw.update(dropout_p=0.3, philox_seed=int(rng.integers(0, 2**62)), philox_offset=int(rng.integers(0, 10**9)))
    okw.update(dropout_p=0.3, philox_seed=kw["philox_seed"], philox_offset=kw["philox_offset"])
  scale = float(rng.choice([D ** -0.5, 0.03, 0.11]))
  plan = {}
  o, lse = hip.forward(q, k, v, bias, causal, scale, plan_out=plan, flags=_FUZZ_FLAGS, num_splits=_FUZZ_SPLITS, **kw)
  assert o.shape == (B, Hq, Nq, D) and lse.shape == (B, Hq, Nq)
  qb, dname = fo.torch_to_bits(q)
  kb, _ = fo.torch_to_bits(k)
  vb, _ = fo.torch_to_bits(v)
  _, o32, lse_ref = fo.oracle_forward(qb, kb, vb, dname, scale=scale, causal=causal, bias=None if bias is None else _f32(bias),
                                      block_keys=plan["block_keys"], **okw)
  got = _f32(o)
  assert np.array_equal(np.isnan(got), np.isnan(o32)), c
  fin = np.isfinite(o32)
  ulp = 2.0 ** -8 if dname == "bf16" else 2.0 ** -11
  # P entries that sit on a rounding boundary may round the other way in the kernel (different fp32 summation
  # order / exp2): one flip moves O by ulp * p_k/l * |v|, so the slack scales with the row's largest probability
  # (these random cases include peaky softmaxes, scale up to 0.11 at D = 1000) — times two for dropout's 1/(1-p).
  g = Hq // Hkv
  sc = (q.float() @ k.float().repeat_interleave(g, 1).transpose(-1, -2)) * scale
  if bias is not None:
    sc = sc + bias.float()
  if causal:
    off = kw.get("causal_offset", Nkv - Nq)
    rows_i = torch.arange(Nq, device="cuda")[:, None]
    cols_i = torch.arange(Nkv, device="cuda")[None, :]
    sc = sc.masked_fill(cols_i > rows_i + off, float("-inf"))
  pmax = torch.exp(sc.max(-1).values - torch.logsumexp(sc, -1)).nan_to_num(0.0).cpu().numpy()[..., None]
  vmax = float(v.float().abs().max())
  flip = 3.0 * ulp * pmax * vmax * (2.0 if c["mode"] == "dropout" else 1.0) + (2.5e-3 if dname == "bf16" else 4e-4)
  err = np.abs(got - o32)
  lim = ulp * np.maximum(np.abs(o32), 2.0 ** -6) + flip
  assert (err[fin] <= np.broadcast_to(lim, err.shape)[fin]).all(), (c, plan, float(err[fin].max()))
  lfin = np.isfinite(lse_ref)
  assert np.array_equal(np.isneginf(_f32(lse)), np.isneginf(lse_ref)), c
  np.testing.assert_allclose(_f32(lse)[lfin], lse_ref[lfin], atol=3e-4, rtol=3e-5, err_msg=str(c))


@pytest.mark.parametrize("seed, fake_cus", [(s_, None) for s_ in range(24)] + [(s_, c_) for s_ in range(24, 32) for c_ in (128, 304)])
def test_randomized_launch_plans_against_oracle(hip, seed, fake_cus, monkeypatch):
  """Long contexts with at most 1.5 rounds of workgroups — where make_plan prices KV splits (under-filled, part of a round, ragged round): whatever count
  it takes, the result equals the unsplit launch to rounding and the oracle on a row sample, tails and the causal flag included.  ``fake_cus``: the same
  walk with the plan pricing a 128- / 304-CU part (FFPA_HIP_FAKE_CUS, test-only: the rates come from the device, the CU count is what another SKU changes
  first) — any plan is a correct launch on any device."""
  rng = np.random.default_rng(7000 + seed)
  cus = torch.cuda.get_device_properties(0).multi_processor_count
  if fake_cus is not None:
    monkeypatch.setenv("FFPA_HIP_FAKE_CUS", str(fake_cus))
    assert hip.load_library().ffpa_attn_query(8) == fake_cus
    cus = fake_cus
  D = int(rng.choice([320, 512, 512, 1024, 448, 640]))
  rows_per_wg = hip.tile_config(hip.padded_head_dim(D))["block_rows"]
  Nq = int(rng.choice([512, 1024, 2048, 4096])) - int(rng.choice([0, 0, 1, 37]))
  nqt = (Nq + rows_per_wg - 1) // rows_per_wg
  want_wgs = int(rng.integers(cus // 8, 3 * cus // 2 + 1))
  H = max(1, want_wgs // nqt)
  group = int(rng.choice([1, 1, 2])) if H % 2 == 0 else 1
  Nkv = int(rng.choice([4096, 8192, 12288, 16384])) - int(rng.choice([0, 0, 5, 100]))
  causal = bool(rng.random() < 0.3) and Nkv >= Nq
  dt = torch.float16 if rng.random() < 0.2 else torch.bfloat16
  q, k, v = _rand((1, H, Nq, D), dt, seed=seed * 3 + 7001), _rand((1, H // group, Nkv, D), dt, seed=seed * 3 + 7002), _rand((1, H // group, Nkv, D), dt, seed=seed * 3 + 7003)
  scale = D ** -0.5
  plan, plan1 = {}, {}
  o, lse = hip.forward(q, k, v, None, causal, scale, plan_out=plan)
  o1, lse1 = hip.forward(q, k, v, None, causal, scale, num_splits=1, plan_out=plan1)
  assert plan1["splits"] == 1 and plan["variant"] == 0, (plan, plan1)
  tag = f"seed {seed}: H{H}/{H // group} Nq{Nq} Nkv{Nkv} D{D} causal {causal} {dt} -> {H * nqt} workgroups, {plan['splits']} splits"
  if plan["splits"] == 1:
    assert torch.equal(o, o1) and torch.equal(lse, lse1), tag
  else:
    assert (o.float() - o1.float()).abs().max().item() <= 4e-3 and (lse - lse1).abs().max().item() <= 1e-4, tag
  heads = sorted(set(int(x) for x in rng.integers(0, H, 2)))
  r0 = int(rng.integers(0, max(1, Nq - 32)))
  for h in heads:
    hs, ks = slice(h, h + 1), slice(h // group, h // group + 1)
    _check_vs_oracle(o[:, hs], lse[:, hs], q[:, hs], k[:, ks], v[:, ks], causal=causal, rows=(r0, min(Nq, r0 + 32)), block_keys=plan["block_keys"],
                     name=tag + f" head {h}", split=plan["splits"] > 1)


# ----------------------------------------------------------------------------- streams and HIP graphs
def test_runs_on_the_callers_stream_and_in_a_hip_graph(hip):
  """The C-ABI launches on the stream it is given (torch's current stream), never synchronises and allocates
  nothing itself, so a warmed-up call can be captured into a HIP graph and replayed (the decode path included)."""
  from ffpa_attn_amd import ffpa_attn_func

  q, k, v = _rand((1, 8, 640, 512), seed=201), _rand((1, 2, 2048, 512), seed=202), _rand((1, 2, 2048, 512), seed=203)
  qd = _rand((4, 8, 1, 512), seed=204)
  kd, vd = _rand((4, 2, 4096, 512), seed=205), _rand((4, 2, 4096, 512), seed=206)
  ref = ffpa_attn_func(q, k, v, is_causal=True, enable_gqa=True)
  refd = ffpa_attn_func(qd, kd, vd, enable_gqa=True)
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):
    out = ffpa_attn_func(q, k, v, is_causal=True, enable_gqa=True)
    outd = ffpa_attn_func(qd, kd, vd, enable_gqa=True)
  side.synchronize()
  assert torch.equal(out, ref) and torch.equal(outd, refd)

  static_q, static_qd = q.clone(), qd.clone()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    go = ffpa_attn_func(static_q, k, v, is_causal=True, enable_gqa=True)
    god = ffpa_attn_func(static_qd, kd, vd, enable_gqa=True)
  static_q.copy_(q * 0.5)
  static_qd.copy_(qd * 0.5)
  g.replay()
  torch.cuda.synchronize()
  assert torch.equal(go, ffpa_attn_func(q * 0.5, k, v, is_causal=True, enable_gqa=True))
  assert torch.equal(god, ffpa_attn_func(qd * 0.5, kd, vd, enable_gqa=True))
