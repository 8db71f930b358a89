"""Packed sequences on the GPU (`pytest -m gpu`): ffpa_attn_varlen_func / ffpa_attn_varlen_fwd — ONE launch for the batch — against

* the pinned CPU oracle, sequence by sequence (the recurrence the dense path is pinned to: tests/test_oracle.py);
* the dense kernel on each sequence alone, BIT FOR BIT (the packed kernel runs the dense kernel's tile text on per-sequence arguments: under
  FFPA_FLAG_DETERMINISTIC — no KV splits, no wide-row tile — a dense launch of one sequence computes the same bits: LSE in both dtypes, O in bf16;
  fp16 O up to the compiler's one-or-two-roundings choice in the very last instruction);
* PyTorch SDPA per sequence at the reference's tolerance;

over the shape list of the reference's own packed-sequence test (tests/test_ffpa_cute_sm100.py:1085-1157: uneven lengths, Nq < Nkv and Nq > Nkv under the
tail-aligned causal mask, zero-length key / query sequences, residues around the row tile, GQA / MQA, 70 sequences with empty ones in between) and its
contract for rows without a visible key: O = 0 and LSE = -inf exactly, with no K / V byte read (:1160-1183)."""

import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_fwd_gpu import TOL, _check_vs_oracle, _close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
  if not torch.cuda.is_available():
    pytest.fail("these tests need a GPU; run with -m 'not gpu' on CPU boxes")
  from ffpa_attn_amd import hip as h

  h.load_library()  # fail loudly if the extension is missing: there is no fallback
  return h


def _cu(lens):
  return torch.tensor([0, *np.cumsum(lens).tolist()], dtype=torch.int32, device="cuda")


def _make(lens_q, lens_k, hq, hkv, d, dtype, seed=0, scale=1.0):
  g = torch.Generator(device="cuda").manual_seed(seed)
  tq, tk = int(sum(lens_q)), int(sum(lens_k))
  q = torch.randn((tq, hq, d), dtype=dtype, device="cuda", generator=g) * scale
  k = torch.randn((tk, hkv, d), dtype=dtype, device="cuda", generator=g) * scale
  v = torch.randn((tk, hkv, d), dtype=dtype, device="cuda", generator=g) * scale
  return q, k, v


def _seq(t, a, b):
  """rows [a, b) of a packed [T, H, D] tensor as the dense layout [1, H, n, D] (a view)"""
  return t[a:b].transpose(0, 1).unsqueeze(0)


def _check_packed(hip, q, k, v, lens_q, lens_k, causal, out, lse, *, oracle=True, dense_bits=True, sdpa=True, name="", max_q=None, max_k=None, split=None):
  """out / lse of one packed call against the three references, sequence by sequence.  (max_q / max_k: what the call announced, if not the longest sequence —
  a launch that split its KV ranges merged fp32 partials: equal to the dense kernel to rounding, so the bit comparison is skipped for it.)"""
  hq, d = q.size(1), q.size(2)
  # (a launch that leaves most of the chip idle splits its KV ranges: fp32 partials + merge — the oracle check then carries the P-rounding noise of several walks)
  if split is None:  # (None: what the library does by itself for this shape class)
    split = hip.varlen_launch_plan(len(lens_q), hq, k.size(1), max(max_q or max(lens_q), 1), max(max_k or max(lens_k), 1), d, dtype=q.dtype, causal=causal, total_q=q.size(0))["splits"] > 1
  dense_bits = dense_bits and not split
  assert out.shape == q.shape and out.dtype == q.dtype and lse.shape == (hq, q.size(0)) and lse.dtype == torch.float32
  bq, bk = np.cumsum([0, *lens_q]), np.cumsum([0, *lens_k])
  scale = 1.0 / math.sqrt(d)
  bc = hip.varlen_launch_plan(len(lens_q), hq, k.size(1), max(max(lens_q), 1), max(max(lens_k), 1), d)["block_keys"]
  for i, (nq, nk) in enumerate(zip(lens_q, lens_k)):
    qs, qe, ks, ke = int(bq[i]), int(bq[i + 1]), int(bk[i]), int(bk[i + 1])
    if nq == 0:
      continue
    o_i, l_i = _seq(out, qs, qe), lse[:, qs:qe].unsqueeze(0)
    dead = nq if nk == 0 else (max(0, nq - nk) if causal else 0)  # rows without a visible key: the first Nq - Nkv of a causal sequence
    if dead:
      assert torch.all(o_i[:, :, :dead] == 0), f"{name} seq {i}: rows without a visible key must be exactly 0"
      assert torch.all(l_i[:, :, :dead] == -float("inf")), f"{name} seq {i}: their LSE must be exactly -inf"
    if dead == nq:
      continue
    assert torch.isfinite(o_i[:, :, dead:].float()).all() and torch.isfinite(l_i[:, :, dead:]).all(), f"{name} seq {i}"
    q_i, k_i, v_i = _seq(q, qs, qe), _seq(k, ks, ke), _seq(v, ks, ke)
    if oracle:
      _check_vs_oracle(o_i, l_i, q_i, k_i, v_i, causal=causal, causal_offset=nk - nq, rows=(dead, nq), block_keys=bc, name=f"{name} seq {i} vs oracle", split=split)
    if dense_bits and nq > 32:  # (dense launches of <= 32 rows run the short-query tiles: another kernel, equal to rounding only)
      o_d, l_d = hip.forward(q_i, k_i, v_i, None, causal, scale, flags=hip.FLAG_DETERMINISTIC)
      assert torch.equal(l_d[:, :, dead:], l_i[:, :, dead:]), f"{name} seq {i}: LSE differs from the dense kernel's bits"
      if q.dtype == torch.bfloat16:
        assert torch.equal(o_d[:, :, dead:], o_i[:, :, dead:]), f"{name} seq {i}: O differs from the dense kernel's bits"
      else:
        # fp16: the accumulators are the same bits; the LAST instruction — O * (1 / l) rounded to fp16 — is hipcc's choice per element between
        # v_fma_mixlo_f16 (one rounding) and v_mul_f32 + v_cvt_f16_f32 (two), and it chooses differently in the two kernels: where the fp32
        # product lands on an fp16 rounding midpoint (~ 2^-13 of the elements) the results differ by one fp16 spacing
        a, b = o_d[:, :, dead:].float(), o_i[:, :, dead:].float()
        diff = (a - b).abs()
        spacing = torch.maximum(a.abs(), b.abs()).clamp_min(2.0 ** -14) * 2.0 ** -10
        assert torch.all(diff <= spacing), f"{name} seq {i}: fp16 O differs from the dense kernel's by more than one spacing ({diff.max().item():.3e})"
        assert (diff > 0).float().mean().item() <= 2e-3, f"{name} seq {i}: {(diff > 0).float().mean().item():.2e} of the fp16 outputs differ from the dense kernel's"
    if sdpa:
      live = slice(dead, nq)
      mask = None
      if causal:
        rows = torch.arange(nq, device="cuda").view(-1, 1)
        cols = torch.arange(nk, device="cuda").view(1, -1)
        mask = (cols <= rows + (nk - nq))[live]
      ref = F.scaled_dot_product_attention(q_i[:, :, live], k_i, v_i, attn_mask=mask, enable_gqa=hq != k.size(1))
      _close(o_i[:, :, live], ref, q.dtype, f"{name} seq {i} vs SDPA")


REFERENCE_SHAPES = [  # tests/test_ffpa_cute_sm100.py:1089-1114 (data: lengths and head counts)
  pytest.param([130, 512, 7, 1024], None, 4, 4, id="uneven_equal"),
  pytest.param([64, 200, 1], [512, 333, 1024], 4, 4, id="cross_sq_lt_sk"),
  pytest.param([512, 700, 300], [64, 129, 1], 4, 4, id="cross_sq_gt_sk"),
  pytest.param([128, 256], [0, 256], 2, 2, id="zero_len_k"),
  pytest.param([0, 256, 130], [64, 256, 130], 2, 2, id="zero_len_q"),
  pytest.param([1, 127, 128, 129, 255], None, 2, 2, id="residue_m"),
  pytest.param([777], None, 2, 2, id="single_seq"),
  pytest.param([255, 1024], None, 8, 2, id="gqa_4to1"),
  pytest.param([300, 129], None, 8, 1, id="mqa_8to1"),
  pytest.param([(0 if i % 7 == 0 else 40 + 9 * i) for i in range(70)], None, 2, 2, id="batch70_with_zeros"),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("lens_q, lens_k, hq, hkv", REFERENCE_SHAPES)
def test_packed_call_matches_oracle_dense_bits_and_sdpa_per_sequence(hip, lens_q, lens_k, hq, hkv, causal, dtype):
  from ffpa_attn_amd import ffpa_attn_varlen_func

  lens_k = lens_q if lens_k is None else lens_k
  d = 512
  q, k, v = _make(lens_q, lens_k, hq, hkv, d, dtype, scale=0.25)  # (the reference test's randn / 4)
  cu_q, cu_k = _cu(lens_q), _cu(lens_k)
  out, lse = ffpa_attn_varlen_func(q, k, v, cu_q, cu_k, max(lens_q), max(lens_k), causal=causal, enable_gqa=hq != hkv, return_lse=True)
  big = len(lens_q) > 8  # (the 70-sequence case: the oracle on a sample of its sequences only — it is a CPU loop)
  _check_packed(hip, q, k, v, lens_q, lens_k, causal, out, lse, oracle=not big, name=f"D{d}")
  if hip.varlen_launch_plan(len(lens_q), hq, hkv, max(lens_q), max(max(lens_k), 1), d, dtype=dtype, causal=causal, total_q=q.size(0))["splits"] > 1:
    # (these few-head batches leave most of the chip idle: the library split their KV ranges.  The one-range launch of the same batch keeps the dense kernel's bits)
    one, one_lse = hip.varlen_forward(q, k, v, cu_q, cu_k, max(lens_q), max(lens_k), causal, d ** -0.5, num_splits=1)
    _check_packed(hip, q, k, v, lens_q, lens_k, causal, one, one_lse, oracle=not big, sdpa=False, split=False, name=f"D{d}, one KV range")
  if big:
    keep = [3, 8, 22, 41, 69]
    for i in keep:
      qs, qe = int(sum(lens_q[:i])), int(sum(lens_q[: i + 1]))
      _check_vs_oracle(_seq(out, qs, qe), lse[:, qs:qe].unsqueeze(0), _seq(q, qs, qe), _seq(k, qs, qe), _seq(v, qs, qe), causal=causal,
                       causal_offset=0, block_keys=64, name=f"batch70 seq {i}")


@pytest.mark.parametrize("d", [64, 72, 128, 192, 256, 320, 384, 448, 576, 640, 768, 1024])
def test_every_head_dim_of_the_packed_kernel(hip, d):
  """every instantiation (128 ... 1024; smaller and in-between head dims run on the next one with their missing columns read as zeros): ragged lengths,
  causal, GQA."""
  lens_q, lens_k = [200, 33, 129, 64], [264, 33, 300, 17]
  for dtype, causal in ((torch.bfloat16, True), (torch.float16, False)):
    q, k, v = _make(lens_q, lens_k, 4, 2, d, dtype, seed=d)
    out, lse = hip.varlen_forward(q, k, v, _cu(lens_q), _cu(lens_k), max(lens_q), max(lens_k), causal, 1.0 / math.sqrt(d))
    # (the dense call serves D = 64 with its 32x32x16 kernel: another mapping, equal to rounding only — from 72 up both run the 16x16x32 tile)
    _check_packed(hip, q, k, v, lens_q, lens_k, causal, out, lse, sdpa=d <= 512, dense_bits=d > 64, name=f"D{d} {dtype} causal={causal}")


def test_head_dim_that_is_not_a_multiple_of_8_is_padded_by_copies(hip):
  lens = [100, 260]
  q, k, v = _make(lens, lens, 2, 2, 100, torch.bfloat16)
  out, lse = hip.varlen_forward(q, k, v, _cu(lens), _cu(lens), 260, 260, True, 0.1)
  assert out.shape == (360, 2, 100) and out.is_contiguous()
  _check_packed(hip, q, k, v, lens, lens, True, out, lse, dense_bits=False, sdpa=False, name="D100")
  for i, (a, b) in enumerate(((0, 100), (100, 360))):
    ref = F.scaled_dot_product_attention(_seq(q, a, b), _seq(k, a, b), _seq(v, a, b), is_causal=True, scale=0.1)
    _close(_seq(out, a, b), ref, torch.bfloat16, f"D100 seq {i}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_rows_without_a_visible_key_read_no_kv(hip, dtype):
  """tests/test_ffpa_cute_sm100.py:1160-1183: every key sequence is empty and the K / V buffers are NaN — O = 0 and LSE = -inf must come from the mask,
  not from arithmetic on data that should never have been loaded.  And the mixed form: one live sequence between two empty ones, NaN rows around its keys."""
  from ffpa_attn_amd import ffpa_attn_varlen_func

  lens_q, d = [130, 256, 1], 512
  cu_q, cu_k = _cu(lens_q), _cu([0, 0, 0])
  q = torch.randn(sum(lens_q), 2, d, device="cuda", dtype=dtype)
  k = torch.full((0, 2, d), float("nan"), device="cuda", dtype=dtype)
  out, lse = ffpa_attn_varlen_func(q, k, k.clone(), cu_q, cu_k, max(lens_q), 1, causal=True, return_lse=True)
  assert torch.all(out == 0) and torch.all(lse == -float("inf"))
  # a live sequence whose neighbours' key rows are NaN: a read past the sequence's own keys would poison it
  lens_k = [64, 200, 64]
  q, k, v = _make(lens_q, lens_k, 2, 2, d, dtype)
  k[:64], k[264:], v[:64], v[264:] = float("nan"), float("nan"), float("nan"), float("nan")
  out, lse = ffpa_attn_varlen_func(q, k, v, cu_q, _cu(lens_k), max(lens_q), max(lens_k), return_lse=True)
  mid = _seq(out, 130, 386)
  assert torch.isfinite(mid.float()).all() and torch.isfinite(lse[:, 130:386]).all()
  ref = F.scaled_dot_product_attention(_seq(q, 130, 386), _seq(k, 64, 264), _seq(v, 64, 264))
  _close(mid, ref, dtype, "live sequence between NaN neighbours")


def test_zero_copy_strided_views_and_oversized_max_seqlen(hip):
  """q / k / v as slices of one packed qkv buffer [T, 3, H, D] (row stride 3 H D) and an output that is written into a dense tensor: nothing is copied
  (strides are part of the call); max_seqlen_q larger than every sequence only adds workgroups that leave at once."""
  lens = [300, 45, 512]
  t, h, d = sum(lens), 4, 320
  g = torch.Generator(device="cuda").manual_seed(5)
  qkv = torch.randn((t, 3, h, d), dtype=torch.bfloat16, device="cuda", generator=g)
  q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
  assert not q.is_contiguous()
  cu = _cu(lens)
  out, lse = hip.varlen_forward(q, k, v, cu, cu, 512, 512, True, d ** -0.5, num_splits=1)
  out2, lse2 = hip.varlen_forward(q.contiguous(), k.contiguous(), v.contiguous(), cu, cu, 4096, 8192, True, d ** -0.5, num_splits=1)
  assert torch.equal(out, out2) and torch.equal(lse, lse2)
  _check_packed(hip, q, k, v, lens, lens, True, out, lse, name="strided")
  # left to the library, this launch (4 heads x 7 row tiles with rows) splits its KV ranges — by the announced maxima, which are part of the shape class: equal to
  # the one-range launch to merge rounding
  for mq, mk in ((512, 512), (4096, 8192)):
    out3, lse3 = hip.varlen_forward(q, k, v, cu, cu, mq, mk, True, d ** -0.5)
    _same_to_merge_rounding(out3, lse3, out, lse, f"strided, max_seqlen {mq} / {mk}")


def test_softmax_scale_conventions_and_exact_recurrence(hip):
  """scale None = 1 / sqrt(D); a negative and a zero scale reach the kernel as (-Q, |scale|) / (Q = 0): as on the dense path; rescale_threshold = 0 is
  the exact recurrence (equal to the lazy one to rounding)."""
  from ffpa_attn_amd import ffpa_attn_varlen_func

  lens = [150, 70]
  q, k, v = _make(lens, lens, 2, 2, 512, torch.bfloat16)
  cu = _cu(lens)
  base = ffpa_attn_varlen_func(q, k, v, cu, None, 150, 150)
  assert torch.equal(base, ffpa_attn_varlen_func(q, k, v, cu, cu, 150, 150, softmax_scale=512 ** -0.5))
  neg = ffpa_attn_varlen_func(q, k, v, cu, cu, 150, 150, softmax_scale=-(512 ** -0.5))
  assert torch.equal(neg, ffpa_attn_varlen_func(-q, k, v, cu, cu, 150, 150))
  zero = ffpa_attn_varlen_func(q, k, v, cu, cu, 150, 150, softmax_scale=0.0)
  for a, b in ((0, 150), (150, 220)):
    _close(_seq(zero, a, b), _seq(v, a, b).float().mean(dim=2, keepdim=True).expand(-1, -1, b - a, -1), torch.bfloat16, "scale 0 = mean of V")
  exact = ffpa_attn_varlen_func(q, k, v, cu, cu, 150, 150, rescale_threshold=0.0)
  _close(exact, base, torch.bfloat16, "exact vs lazy recurrence")


def test_packed_call_captures_into_a_hip_graph_and_follows_the_device_side_boundaries(hip):
  """No host read, no synchronisation: the call captures as it is, and a replay follows cu_seqlens written IN PLACE (same totals, other boundaries)."""
  lens_a, lens_b = [100, 300, 112], [256, 0, 256]
  q, k, v = _make(lens_a, lens_a, 4, 4, 512, torch.bfloat16)
  cu = _cu(lens_a)
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):
    hip.varlen_forward(q, k, v, cu, cu, 512, 512, True, 512 ** -0.5)  # (warm-up: the kernel attribute is set outside the capture)
  torch.cuda.current_stream().wait_stream(side)
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    out, lse = hip.varlen_forward(q, k, v, cu, cu, 512, 512, True, 512 ** -0.5)
  for lens in (lens_a, lens_b, lens_a):
    cu.copy_(_cu(lens))
    g.replay()
    torch.cuda.synchronize()
    eager, eager_lse = hip.varlen_forward(q, k, v, cu, cu, 512, 512, True, 512 ** -0.5)
    assert torch.equal(out, eager) and torch.equal(lse, eager_lse)
    _check_packed(hip, q, k, v, lens, lens, True, out, lse, oracle=False, name=f"replay {lens}")


def test_backward_of_the_packed_call_matches_per_sequence_sdpa():
  """tests/test_ffpa_cute_sm80.py:269-…: autograd through the packed call against SDPA per sequence; here the backward is the dense path's SDPA-backward
  hookup per sequence on the saved O / LSE.  Includes a causal sequence with more queries than keys (its first rows get zero gradients)."""
  from ffpa_attn_amd import ffpa_attn_varlen_func

  lens_q, lens_k, h, d = [96, 40, 130], [96, 72, 50], 2, 320
  q, k, v = _make(lens_q, lens_k, h, h, d, torch.bfloat16, scale=0.5)
  for causal in (False, True):
    qa, ka, va = (t.clone().requires_grad_(True) for t in (q, k, v))
    out = ffpa_attn_varlen_func(qa, ka, va, _cu(lens_q), _cu(lens_k), max(lens_q), max(lens_k), causal=causal)
    go = torch.randn_like(out)
    out.backward(go)
    qr, kr, vr = (t.clone().float().requires_grad_(True) for t in (q, k, v))
    bq, bk = np.cumsum([0, *lens_q]), np.cumsum([0, *lens_k])
    refs = []
    for i, (nq, nk) in enumerate(zip(lens_q, lens_k)):
      qs, qe, ks, ke = int(bq[i]), int(bq[i + 1]), int(bk[i]), int(bk[i + 1])
      mask = None
      if causal:
        mask = torch.arange(nk, device="cuda").view(1, -1) <= torch.arange(nq, device="cuda").view(-1, 1) + (nk - nq)
      o = F.scaled_dot_product_attention(_seq(qr, qs, qe), _seq(kr, ks, ke), _seq(vr, ks, ke), attn_mask=mask)
      o = torch.nan_to_num(o, nan=0.0)  # rows without a visible key: the packed contract is O = 0 (their gradients vanish)
      refs.append(o[0].transpose(0, 1))
    ref = torch.cat(refs, dim=0)
    _close(out, ref.to(out.dtype), torch.bfloat16, f"forward causal={causal}")
    ref.backward(go.float())
    for name, got, want in (("dq", qa.grad, qr.grad), ("dk", ka.grad, kr.grad), ("dv", va.grad, vr.grad)):
      want = torch.nan_to_num(want, nan=0.0)
      err = (got.float() - want).abs().max().item()
      assert err <= 3e-2 * max(1.0, want.abs().max().item()), f"{name} causal={causal}: {err:.3e}"


def test_op_registration_of_the_packed_call(hip):
  """ffpa_attn::_varlen_fwd_hip through torch.library's own checks (schema, fake implementation, dispatch) and through torch.compile."""
  from ffpa_attn_amd import ffpa_attn_varlen_func

  lens = [70, 130]
  q, k, v = _make(lens, lens, 2, 2, 320, torch.bfloat16)
  cu = _cu(lens)
  torch.library.opcheck(torch.ops.ffpa_attn._varlen_fwd_hip.default, (q, k, v, cu, cu, 130, 130, 320 ** -0.5, 1),
                        test_utils=("test_schema", "test_faketensor"))
  eager = ffpa_attn_varlen_func(q, k, v, cu, cu, 130, 130, causal=True)
  compiled = torch.compile(lambda a, b, c: ffpa_attn_varlen_func(a, b, c, cu, cu, 130, 130, causal=True) * 2.0)(q, k, v)
  assert torch.equal(compiled, eager * 2.0)


def test_c_abi_of_the_packed_call_on_raw_pointers(hip):
  """The boundary itself: ffpa_attn_varlen_fwd on a hand-filled parameter block (no torch types behind the pointers' back), stream given explicitly."""
  import ctypes

  lens = [190, 66]
  q, k, v = _make(lens, lens, 4, 1, 256, torch.float16)
  cu = _cu(lens)
  o = torch.full_like(q, float("nan"))
  lse = torch.full((4, 256), float("nan"), dtype=torch.float32, device="cuda")
  lib = hip.load_library()
  p = hip.FfpaVarlenFwdParams()
  p.struct_size, p.abi_version = ctypes.sizeof(hip.FfpaVarlenFwdParams), hip.ABI_VERSION
  p.q, p.k, p.v, p.o, p.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr()
  p.cu_seqlens_q = p.cu_seqlens_kv = cu.data_ptr()
  p.batch, p.heads_q, p.heads_kv, p.head_dim, p.max_seqlen_q, p.max_seqlen_kv = 2, 4, 1, 256, 190, 190
  p.q_stride[:], p.k_stride[:], p.v_stride[:], p.o_stride[:] = [4 * 256, 256], [256, 256], [256, 256], [4 * 256, 256]
  p.lse_stride_head, p.dtype, p.causal, p.softmax_scale, p.rescale_threshold = 256, 1, 1, 1.0 / 16, -1.0
  s = torch.cuda.Stream()
  s.wait_stream(torch.cuda.current_stream())
  assert lib.ffpa_attn_varlen_fwd(ctypes.byref(p), ctypes.c_void_p(s.cuda_stream)) == 0, lib.ffpa_attn_last_error()
  s.synchronize()
  _check_packed(hip, q, k, v, lens, lens, True, o, lse, name="C-ABI")


def test_c_abi_kv_splits_stay_inside_the_scratch_the_caller_sized(hip):
  """ABI 6 on raw pointers: the caller's scratch (ffpa_attn_varlen_fwd_workspace_bytes for ITS total_q) is all a split launch writes — a parameter block whose total_q
  is smaller than the boundaries on the device say (a caller's bug the host side cannot see) loses the sequences past it, not memory: the words behind the scratch
  keep their pattern, the sequences inside total_q are right; with the honest total_q the same call is whole."""
  import ctypes

  lens_q, lens_k = [3, 2, 4, 3], [900, 1500, 700, 1100]
  q, k, v = _make(lens_q, lens_k, 8, 2, 512, torch.bfloat16, seed=5)
  cu_q, cu_k = _cu(lens_q), _cu(lens_k)
  lib = hip.load_library()
  ref, ref_lse = hip.varlen_forward(q, k, v, cu_q, cu_k, 4, 1500, True, 512 ** -0.5, num_splits=1)
  for total_q in (12, 5):
    o = torch.full_like(q, float("nan"))
    lse = torch.full((8, 12), float("nan"), dtype=torch.float32, device="cuda")
    p = hip.FfpaVarlenFwdParams()
    p.struct_size, p.abi_version = ctypes.sizeof(hip.FfpaVarlenFwdParams), hip.ABI_VERSION
    p.q, p.k, p.v, p.o, p.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr()
    p.cu_seqlens_q, p.cu_seqlens_kv = cu_q.data_ptr(), cu_k.data_ptr()
    p.batch, p.heads_q, p.heads_kv, p.head_dim, p.max_seqlen_q, p.max_seqlen_kv = 4, 8, 2, 512, 4, 1500
    p.q_stride[:], p.k_stride[:], p.v_stride[:], p.o_stride[:] = [8 * 512, 512], [2 * 512, 512], [2 * 512, 512], [8 * 512, 512]
    p.lse_stride_head, p.dtype, p.causal, p.softmax_scale, p.rescale_threshold = 12, 0, 1, 512 ** -0.5, -1.0
    p.total_q, p.num_splits, p.flags = total_q, 5, hip.FLAG_FORCE_SPLITS
    want = lib.ffpa_attn_varlen_fwd_workspace_bytes(ctypes.byref(p))
    assert want == 5 * 8 * total_q * (512 + 1) * 4
    guard = 1 << 20
    ws = torch.full(((want + guard) // 4,), 1.5, dtype=torch.float32, device="cuda")
    p.workspace, p.workspace_bytes = ws.data_ptr(), want
    plan = (ctypes.c_int * 5)()
    assert lib.ffpa_attn_varlen_fwd_plan(ctypes.byref(p), plan) == 0 and plan[4] == 5 and plan[3] == 4 * 2 * 5, list(plan)
    assert lib.ffpa_attn_varlen_fwd(ctypes.byref(p), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0, lib.ffpa_attn_last_error()
    torch.cuda.synchronize()
    assert torch.all(ws[want // 4:] == 1.5), f"total_q = {total_q}: the launch wrote behind the scratch"
    rows = 12 if total_q == 12 else 5  # (sequences 0 and 1 end at row 5; sequence 2 — rows 5 .. 9 — does not fit a scratch of 5 rows per head)
    _same_to_merge_rounding(o[:rows], lse[:, :rows].contiguous(), ref[:rows], ref_lse[:, :rows].contiguous(), f"C-ABI splits, total_q = {total_q}")
    if total_q == 5:
      assert torch.isnan(o[5:].float()).all() and torch.isnan(lse[:, 5:]).all(), "rows past total_q must not be written"


def test_randomized_packed_batches(hip):
  """Random batches (1 ... 12 sequences of 0 ... 700 tokens, independent query / key lengths, MHA / GQA / MQA, causal or not, both dtypes, a head dim of every
  tile family) against the dense kernel's bits and SDPA per sequence, and the oracle on one sequence of each batch.  FFPA_VARLEN_FUZZ=a:b picks the seeds."""
  import os

  lo, hi = (int(x) for x in os.environ.get("FFPA_VARLEN_FUZZ", "0:48").split(":"))
  for seed in range(lo, hi):
    rng = np.random.default_rng(1000 + seed)
    nseq = int(rng.integers(1, 13))
    lens_q = [int(x) for x in rng.choice([0, 1, 7, 31, 33, 64, 127, 128, 129, 200, 333, 512, 700], size=nseq)]
    lens_k = lens_q if rng.random() < 0.4 else [int(x) for x in rng.choice([0, 1, 17, 64, 65, 128, 255, 256, 300, 640], size=nseq)]
    if sum(lens_q) == 0:
      lens_q[0] = 40
    hkv = int(rng.choice([1, 2, 4]))
    hq = hkv * int(rng.choice([1, 2, 4]))
    d = int(rng.choice([128, 256, 320, 512, 640, 1024]))
    dtype = torch.bfloat16 if rng.random() < 0.6 else torch.float16
    causal = bool(rng.random() < 0.5)
    q, k, v = _make(lens_q, lens_k, hq, hkv, d, dtype, seed=seed)
    mq, mk = max(lens_q), max(max(lens_k), 1)
    mq_said = mq + int(rng.integers(0, 200))
    # FFPA_VARLEN_FUZZ_SPLITS=n: every launch splits its KV ranges in n (FLAG_FORCE_SPLITS: prefill launches too — every row tile its own visible tiles; as many as the longest sequence has tiles)
    forced = int(os.environ.get("FFPA_VARLEN_FUZZ_SPLITS", "0"))
    plan = {}
    out, lse = hip.varlen_forward(q, k, v, _cu(lens_q), _cu(lens_k), mq_said, mk, causal, 1.0 / math.sqrt(d), plan_out=plan,
                                  **(dict(num_splits=forced, flags=hip.FLAG_FORCE_SPLITS) if forced > 1 else {}))
    name = f"fuzz seed {seed}: lens_q={lens_q} lens_k={lens_k} Hq={hq} Hkv={hkv} D={d} {dtype} causal={causal}"
    was_split = plan["splits"] > 1
    _check_packed(hip, q, k, v, lens_q, lens_k, causal, out, lse, oracle=False, sdpa=d <= 512, name=name, max_q=mq_said, max_k=mk, split=was_split)
    live = [i for i, (a, b) in enumerate(zip(lens_q, lens_k)) if a > 0 and b > 0 and not (causal and a > b)]
    if live:
      i = live[seed % len(live)]
      qs, ks = int(sum(lens_q[:i])), int(sum(lens_k[:i]))
      bc = hip.varlen_launch_plan(nseq, hq, hkv, mq, mk, d)["block_keys"]
      _check_vs_oracle(_seq(out, qs, qs + lens_q[i]), lse[:, qs:qs + lens_q[i]].unsqueeze(0), _seq(q, qs, qs + lens_q[i]), _seq(k, ks, ks + lens_k[i]),
                       _seq(v, ks, ks + lens_k[i]), causal=causal, causal_offset=lens_k[i] - lens_q[i], block_keys=bc, name=name + f" seq {i} vs oracle", split=was_split)


@pytest.mark.parametrize("hq, hkv, d, dtype", [(32, 8, 512, torch.bfloat16), (16, 2, 320, torch.float16), (8, 1, 1024, torch.bfloat16), (24, 8, 128, torch.bfloat16)])
def test_decode_batches_pack_the_heads_of_a_kv_group_into_rows(hip, hq, hkv, d, dtype):
  """One query token per sequence (continuous-batching decode) under GQA: the launch runs one workgroup per (sequence, KV head) with the group's query heads
  as the rows of its tile (the reference's pack_gqa) — per row the same arithmetic: bit-identical to the unpacked launch (FLAG_NO_PACK_GQA), O and LSE; against
  SDPA per sequence; sequences without a token or without keys in between."""
  lens_k = [700, 0, 64, 1300, 129, 2048, 1, 333]
  lens_q = [1, 1, 0, 1, 1, 1, 1, 1]
  q, k, v = _make(lens_q, lens_k, hq, hkv, d, dtype, seed=hq + d)
  cu_q, cu_k = _cu(lens_q), _cu(lens_k)
  plan = {}
  out, lse = hip.varlen_forward(q, k, v, cu_q, cu_k, 1, max(lens_k), True, d ** -0.5, plan_out=plan, num_splits=1)  # (one KV range per sequence: the bits below)
  assert plan["workgroups"] == len(lens_q) * hkv and "packed into rows" in plan["kernel"], plan
  plan_u = {}
  out_u, lse_u = hip.varlen_forward(q, k, v, cu_q, cu_k, 1, max(lens_k), True, d ** -0.5, flags=hip.FLAG_NO_PACK_GQA, plan_out=plan_u, num_splits=1)
  assert plan_u["workgroups"] == len(lens_q) * hq
  assert torch.equal(out, out_u) and torch.equal(lse, lse_u)
  _check_packed(hip, q, k, v, lens_q, lens_k, True, out, lse, oracle=False, dense_bits=False, sdpa=d <= 512, name=f"packed decode Hq{hq}/Hkv{hkv} D{d}")
  # the oracle on the longest sequence
  i = 5
  qs, ks = sum(lens_q[:i]), sum(lens_k[:i])
  _check_vs_oracle(_seq(out, qs, qs + 1), lse[:, qs:qs + 1].unsqueeze(0), _seq(q, qs, qs + 1), _seq(k, ks, ks + lens_k[i]), _seq(v, ks, ks + lens_k[i]),
                   causal=True, causal_offset=lens_k[i] - 1, block_keys=plan["block_keys"], name="packed decode vs oracle")


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("hq, hkv, d, dtype, nq", [(32, 8, 512, torch.bfloat16, 16), (16, 2, 320, torch.float16, 5), (8, 2, 1024, torch.bfloat16, 16), (24, 8, 128, torch.bfloat16, 42),
                                                   (8, 4, 512, torch.bfloat16, 64), (6, 1, 256, torch.float16, 2)])
def test_short_sequences_pack_heads_and_tokens_into_the_rows_of_one_tile(hip, hq, hkv, d, dtype, nq, causal):
  """A few query tokens per sequence under GQA (speculative decoding, multi-token prediction, small prefill chunks): group x max_seqlen_q rows fit one tile, and the
  launch runs one workgroup per (sequence, KV head) whose rows are (head of the group, token), head-major — per row the same arithmetic: bit-identical to the unpacked
  launch (FLAG_NO_PACK_GQA), O and LSE; ragged token counts (1 ... Nq, none), more tokens than keys under the causal flag (rows without a visible key), no keys."""
  lens_k = [700, 0, 64, 1300, 129, 2048, 1, 333, 3]
  lens_q = [nq, nq, 0, max(1, nq - 1), 1, nq, min(nq, 3), max(1, nq // 2), nq]
  q, k, v = _make(lens_q, lens_k, hq, hkv, d, dtype, seed=hq + d + nq)
  cu_q, cu_k = _cu(lens_q), _cu(lens_k)
  plan, plan_u = {}, {}
  out, lse = hip.varlen_forward(q, k, v, cu_q, cu_k, nq, max(lens_k), causal, d ** -0.5, plan_out=plan, num_splits=1)
  assert plan["workgroups"] == len(lens_q) * hkv and "packed into rows" in plan["kernel"], plan
  out_u, lse_u = hip.varlen_forward(q, k, v, cu_q, cu_k, nq, max(lens_k), causal, d ** -0.5, flags=hip.FLAG_NO_PACK_GQA, plan_out=plan_u, num_splits=1)
  assert plan_u["workgroups"] == len(lens_q) * hq and "packed into rows" not in plan_u["kernel"], plan_u
  assert torch.equal(lse, lse_u), (lse - lse_u).abs().max()
  assert torch.equal(out, out_u)
  _check_packed(hip, q, k, v, lens_q, lens_k, causal, out, lse, oracle=False, dense_bits=False, sdpa=d <= 512, name=f"packed tokens Hq{hq}/Hkv{hkv} D{d} Nq{nq}")
  i = 5
  qs, ks = sum(lens_q[:i]), sum(lens_k[:i])
  _check_vs_oracle(_seq(out, qs, qs + nq), lse[:, qs:qs + nq].unsqueeze(0), _seq(q, qs, qs + nq), _seq(k, ks, ks + lens_k[i]), _seq(v, ks, ks + lens_k[i]),
                   causal=causal, causal_offset=lens_k[i] - nq, block_keys=plan["block_keys"], name="packed tokens vs oracle")
  # ... and with its KV ranges split (the library's own count for this batch): to merge rounding
  ps = {}
  out_s, lse_s = hip.varlen_forward(q, k, v, cu_q, cu_k, nq, max(lens_k), causal, d ** -0.5, plan_out=ps)
  assert ps["splits"] > 1 and "packed into rows" in ps["kernel"], ps
  _same_to_merge_rounding(out_s, lse_s, out, lse, f"packed tokens + {ps['splits']} KV ranges")


def test_randomized_short_query_batches(hip):
  """Random decode-like batches — 1 ... 24 sequences of 0 ... 40 query tokens against 0 ... 6000 keys, MHA / GQA / MQA, a head dim of every tile family, causal or
  not, optional device-side KV lengths over a roomier cache — through every launch form the plan can take (rows packed or per head, one KV range or the library's
  count or a forced one, the non-temporal fetch on or off): the packed / NT forms bit-identical to the plain one at one KV range, split launches equal to it to merge
  rounding, all of it against SDPA per sequence.  FFPA_VARLEN_FUZZ_SHORT=a:b picks the seeds."""
  import os

  lo, hi = (int(x) for x in os.environ.get("FFPA_VARLEN_FUZZ_SHORT", "0:40").split(":"))
  for seed in range(lo, hi):
    rng = np.random.default_rng(7000 + seed)
    nseq = int(rng.integers(1, 25))
    mq = int(rng.choice([1, 1, 1, 2, 3, 4, 8, 16, 40]))
    lens_q = [int(rng.integers(0, mq + 1)) if rng.random() < 0.3 else mq for _ in range(nseq)]
    lens_k = [int(x) for x in rng.choice([0, 1, 17, 64, 65, 300, 640, 1500, 3000, 6000], size=nseq)]
    if sum(lens_q) == 0:
      lens_q[0] = mq
    hkv = int(rng.choice([1, 2, 4]))
    hq = hkv * int(rng.choice([1, 2, 4, 8]))
    d = int(rng.choice([128, 256, 320, 512, 640, 1024]))
    dtype = torch.bfloat16 if rng.random() < 0.6 else torch.float16
    causal = bool(rng.random() < 0.6)
    q, k, v = _make(lens_q, lens_k, hq, hkv, d, dtype, seed=seed)
    cu_q, cu_k, mk = _cu(lens_q), _cu(lens_k), max(max(lens_k), 1)
    used = None
    if rng.random() < 0.3 and sum(lens_k) > 0:
      # a roomier cache: every sequence's range holds `slack` more rows than it uses (NaN: nothing may read them)
      slack = int(rng.integers(1, 200))
      kk = torch.full((sum(lens_k) + slack * nseq, hkv, d), float("nan"), dtype=dtype, device="cuda")
      vv = kk.clone()
      roomy, pos, src = [0], 0, 0
      for n in lens_k:
        kk[pos:pos + n], vv[pos:pos + n] = k[src:src + n], v[src:src + n]
        pos, src = pos + n + slack, src + n
        roomy.append(pos)
      used = torch.tensor(lens_k, dtype=torch.int32, device="cuda")
      k_call, v_call, cu_call, mk_call = kk, vv, torch.tensor(roomy, dtype=torch.int32, device="cuda"), mk + slack
    else:
      k_call, v_call, cu_call, mk_call = k, v, cu_k, mk
    name = f"short fuzz seed {seed}: lens_q={lens_q} lens_k={lens_k} Hq={hq} Hkv={hkv} D={d} {dtype} causal={causal} seqused={used is not None}"

    def run(**kw):
      return hip.varlen_forward(q, k_call, v_call, cu_q, cu_call, mq, mk_call, causal, 1.0 / math.sqrt(d), seqused_k=used, **kw)

    base, base_lse = run(num_splits=1, flags=hip.FLAG_NO_PACK_GQA | hip.FLAG_NO_KV_STREAM)
    _check_packed(hip, q, k, v, lens_q, lens_k, causal, base, base_lse, oracle=False, dense_bits=False, sdpa=d <= 512, name=name)
    for flags in (0, hip.FLAG_KV_STREAM, hip.FLAG_NO_PACK_GQA | hip.FLAG_KV_STREAM):
      o, l = run(num_splits=1, flags=flags)
      assert torch.equal(o, base) and torch.equal(l, base_lse), f"{name}: flags {flags:#x} at one KV range differ from the plain launch"
    plan = {}
    o, l = run(plan_out=plan)
    if plan["splits"] > 1:
      _same_to_merge_rounding(o, l, base, base_lse, f"{name}: the library's {plan['splits']} KV ranges")
    else:
      assert torch.equal(o, base) and torch.equal(l, base_lse), name
    forced = int(rng.choice([2, 3, 5, 8, 13]))
    plan = {}
    o, l = run(num_splits=forced, flags=hip.FLAG_FORCE_SPLITS | (hip.FLAG_NO_PACK_GQA if rng.random() < 0.3 else 0), plan_out=plan)
    if plan["row_tiles"] == 1 and plan["splits"] > 1:
      _same_to_merge_rounding(o, l, base, base_lse, f"{name}: {plan['splits']} forced KV ranges")


def _same_to_merge_rounding(out, lse, ref, ref_lse, name):
  """A KV-split launch against the one-range launch of the same batch: the same softmax, combined from normalised fp32 partials.  Not the same bits: every range
  runs the recurrence from ITS first tile (its own stale row max under the lazy-rescale convention), so P = exp2(s - m) is rounded to 16 bits at another position
  of its binade — the P-rounding noise of two runs, not of one (per run ~ 2^-9 |v| / sqrt(effective keys) rms per element, the row's largest |O| ~ 2.5 / sqrt(effective
  keys); the worst of 10^5 elements at 4 sigma of the difference of two runs: 2^-8 x 1.1): per element one output spacing + 2^-7 (bf16) / 2^-10 (fp16) of the row's largest |O|.  LSE within fp32
  rounding of the exp / log pair; rows without a visible key exactly 0 / -inf in both."""
  dead = ref_lse == -float("inf")
  assert torch.equal(lse == -float("inf"), dead), name
  assert torch.allclose(lse[~dead], ref_lse[~dead], rtol=0, atol=2e-5), f"{name}: LSE differs by {(lse[~dead] - ref_lse[~dead]).abs().max().item():.3e}"
  a, b = out.float(), ref.float()
  assert torch.isfinite(a).all(), name
  bf = out.dtype == torch.bfloat16
  spacing = torch.maximum(a.abs(), b.abs()).clamp_min(2.0 ** -14) * (2.0 ** -7 if bf else 2.0 ** -10)
  allow = spacing + b.abs().amax(dim=-1, keepdim=True) * (2.0 ** -7 if bf else 2.0 ** -10)
  diff = (a - b).abs()
  assert torch.all(diff <= allow), f"{name}: O differs from the one-range launch by {(diff / allow).max().item():.2f} x the allowance"
  assert (diff / allow).mean().item() < 0.1, f"{name}: mean difference {(diff / allow).mean().item():.3f} of the allowance"
  assert torch.all(a[dead.t().unsqueeze(-1).expand_as(a)] == 0), name


@pytest.mark.parametrize("hq, hkv, d, dtype, nq, splits", [
  (32, 8, 512, torch.bfloat16, 1, 0), (32, 8, 512, torch.bfloat16, 1, 3), (8, 8, 320, torch.float16, 1, 5), (4, 1, 1024, torch.bfloat16, 1, 16), (16, 4, 128, torch.bfloat16, 1, 2),
  (8, 8, 512, torch.bfloat16, 37, 4), (8, 2, 256, torch.float16, 100, 7), (8, 2, 640, torch.bfloat16, 64, 0), (4, 4, 512, torch.bfloat16, 128, 64),
  (32, 8, 512, torch.bfloat16, 16, 5), (8, 2, 1024, torch.float16, 9, 0),
  (8, 2, 72, torch.bfloat16, 1, 3), (8, 2, 200, torch.float16, 4, 0), (4, 4, 328, torch.bfloat16, 20, 6),  # head dims between the kernels' (128 / 256 / 384 run them: partial rows of the KERNEL's width)
])
def test_kv_splits_inside_the_packed_launch(hip, hq, hkv, d, dtype, nq, splits):
  """Batches of one row tile per (sequence, head) that leave most of the chip idle split every sequence's KV range over several workgroups — each sequence by ITS OWN
  length, read on the device — and merge fp32 partials in a second kernel of the same call.  Left to the library (splits = 0) and forced (FLAG_FORCE_SPLITS: 2 ... 64
  ranges, more ranges than some sequences have KV tiles; ranges without a key; sequences without a key or a token) against the one-range launch of the same batch (to
  merge rounding), SDPA per sequence and the oracle on the longest sequence; the empty-row contract holds through the merge."""
  lens_k = [2900, 0, 64, 1300, 129, 4096, 1, 33, 640]  # (33 < Nq: the first Nq - 33 rows of that sequence see no key under the causal flag)
  lens_q = [nq, nq, 0, nq, max(1, nq // 2), nq, 1, nq, nq]
  q, k, v = _make(lens_q, lens_k, hq, hkv, d, dtype, seed=hq + d + nq)
  cu_q, cu_k = _cu(lens_q), _cu(lens_k)
  p1, ps = {}, {}
  ref, ref_lse = hip.varlen_forward(q, k, v, cu_q, cu_k, nq, max(lens_k), True, d ** -0.5, num_splits=1, plan_out=p1)
  out, lse = hip.varlen_forward(q, k, v, cu_q, cu_k, nq, max(lens_k), True, d ** -0.5, num_splits=splits, flags=hip.FLAG_FORCE_SPLITS if splits else 0, plan_out=ps)
  assert p1["splits"] == 1 and "merge" not in p1["kernel"]
  tiles = -(-max(lens_k) // ps["block_keys"])
  if not splits and ps["splits"] == 1:
    pytest.skip("left to the library, this batch does not split on this device's CU count (the rule's tables: tests/test_varlen.py)")
  assert ps["splits"] == (min(splits, tiles) if splits else ps["splits"]) and ps["splits"] > 1 and ps["workgroups"] == p1["workgroups"] * ps["splits"], (p1, ps)
  assert ps["kernel"].endswith("+ ffpa_varlen_merge_kernel"), ps
  _same_to_merge_rounding(out, lse, ref, ref_lse, f"splits {ps['splits']} Hq{hq}/Hkv{hkv} D{d} Nq{nq}")
  _check_packed(hip, q, k, v, lens_q, lens_k, True, out, lse, oracle=False, dense_bits=False, sdpa=d <= 512, name=f"KV splits {ps['splits']} Hq{hq}/Hkv{hkv} D{d} Nq{nq}")
  i = 5
  qs, ks = sum(lens_q[:i]), sum(lens_k[:i])
  _check_vs_oracle(_seq(out, qs, qs + nq), lse[:, qs:qs + nq].unsqueeze(0), _seq(q, qs, qs + nq), _seq(k, ks, ks + lens_k[i]), _seq(v, ks, ks + lens_k[i]),
                   causal=True, causal_offset=lens_k[i] - nq, block_keys=ps["block_keys"], name="KV splits vs oracle")
  # the deterministic flag keeps one range per sequence: the one-range launch's bits
  det, det_lse = hip.varlen_forward(q, k, v, cu_q, cu_k, nq, max(lens_k), True, d ** -0.5, flags=hip.FLAG_DETERMINISTIC)
  assert torch.equal(det, ref) and torch.equal(det_lse, ref_lse)


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("lens, hq, hkv, d, dtype", [
  ([4864, 256, 0, 130, 130, 1, 64, 7], 8, 2, 512, torch.bfloat16),                # one long sequence sizes the full grid: 8 x 38 row-tile slots per head, 43 + 8 in the compact one
  ([(0 if i % 9 == 0 else 1 + (37 * i) % 300) for i in range(150)] + [2000], 2, 2, 320, torch.float16),  # 151 sequences: three steps of the 64-sequence scan
  ([2000, 64, 64, 64, 64, 64, 64, 64, 1], 4, 1, 1024, torch.bfloat16),             # 64-row tiles
])
def test_compact_grid_of_ragged_prefill_batches(hip, lens, hq, hkv, d, dtype, causal):
  """A caller that says how many rows q has lets a ragged prefill batch size its grid by the rows there are (ceil(total_q / block rows) + batch row-tile slots per
  head; the kernel finds a slot's (sequence, row tile) on the device) instead of batch x the longest sequence's row tiles: fewer workgroups, the same order, the
  same bits as the full grid (FLAG_NO_COMPACT_GRID) — with one KV range and with forced ranges; checked against the dense kernel's bits / SDPA per sequence."""
  q, k, v = _make(lens, lens, hq, hkv, d, dtype, seed=len(lens) + d)
  cu = _cu(lens)
  for kw in (dict(num_splits=1), dict(num_splits=3, flags=hip.FLAG_FORCE_SPLITS)):
    pc, pf = {}, {}
    out, lse = hip.varlen_forward(q, k, v, cu, cu, max(lens), max(lens), causal, d ** -0.5, plan_out=pc, **kw)
    kw_full = dict(kw, flags=kw.get("flags", 0) | hip.FLAG_NO_COMPACT_GRID)
    full, full_lse = hip.varlen_forward(q, k, v, cu, cu, max(lens), max(lens), causal, d ** -0.5, plan_out=pf, **kw_full)
    br = pc["block_rows"]
    slots = -(-sum(lens) // br) + len(lens)
    assert pc["workgroups"] == slots * hq * pc["splits"] and pf["workgroups"] == len(lens) * pf["row_tiles"] * hq * pf["splits"] and pc["workgroups"] < pf["workgroups"], (pc, pf)
    assert torch.equal(out, full) and torch.equal(lse, full_lse)
  out, lse = hip.varlen_forward(q, k, v, cu, cu, max(lens), max(lens), causal, d ** -0.5, num_splits=1)
  _check_packed(hip, q, k, v, lens, lens, causal, out, lse, oracle=False, sdpa=d <= 512, split=False, name=f"compact grid {len(lens)} sequences D{d}")


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("hq, hkv, d, dtype, splits", [
  (4, 4, 512, torch.bfloat16, 0), (4, 4, 512, torch.bfloat16, 3), (8, 2, 320, torch.float16, 5), (2, 1, 1024, torch.bfloat16, 4), (4, 2, 128, torch.bfloat16, 7), (2, 2, 200, torch.float16, 2),
  (2, 2, 512, torch.bfloat16, 64),
])
def test_kv_splits_of_under_filled_prefill_launches(hip, hq, hkv, d, dtype, splits, causal):
  """PREFILL launches (several row tiles per head) that leave most of the chip idle — a chunk of a long prompt with a few heads per GPU — split the KV range of every
  row tile: under the causal flag the tiles up to ITS diagonal (the kernel restates the tile text's clamp), each sequence by its own lengths.  Left to the library
  (splits = 0) and forced (more ranges than the short row tiles have KV tiles: empty ranges; sequences with more rows than keys: rows without a visible key; a
  sequence without keys) against the one-range launch (to merge rounding), SDPA per sequence and the oracle on one sequence."""
  lens_q = [700, 130, 0, 300, 64, 513]
  lens_k = [5000, 130, 77, 100, 2048, 0 if splits == 3 else 1900]
  q, k, v = _make(lens_q, lens_k, hq, hkv, d, dtype, seed=hq + d + splits)
  cu_q, cu_k = _cu(lens_q), _cu(lens_k)
  p1, ps = {}, {}
  ref, ref_lse = hip.varlen_forward(q, k, v, cu_q, cu_k, max(lens_q), max(lens_k), causal, d ** -0.5, num_splits=1, plan_out=p1)
  out, lse = hip.varlen_forward(q, k, v, cu_q, cu_k, max(lens_q), max(lens_k), causal, d ** -0.5, num_splits=splits, flags=hip.FLAG_FORCE_SPLITS if splits else 0, plan_out=ps)
  assert p1["splits"] == 1 and "merge" not in p1["kernel"] and p1["row_tiles"] > 1
  tiles = -(-max(lens_k) // ps["block_keys"])
  if not splits and ps["splits"] == 1:
    pytest.skip("left to the library, this batch does not split on this device's CU count (the rule's tables: tests/test_varlen.py)")
  assert ps["splits"] == (min(splits, tiles) if splits else ps["splits"]) and ps["splits"] > 1 and ps["workgroups"] == p1["workgroups"] * ps["splits"], (p1, ps)
  assert ps["kernel"].endswith("+ ffpa_varlen_merge_kernel"), ps
  name = f"prefill KV splits {ps['splits']} Hq{hq}/Hkv{hkv} D{d} {'causal' if causal else 'full'}"
  _same_to_merge_rounding(out, lse, ref, ref_lse, name)
  _check_packed(hip, q, k, v, lens_q, lens_k, causal, out, lse, oracle=False, dense_bits=False, sdpa=d <= 512, name=name)
  i = 0
  _check_vs_oracle(_seq(out, 0, lens_q[i]), lse[:, :lens_q[i]].unsqueeze(0), _seq(q, 0, lens_q[i]), _seq(k, 0, lens_k[i]), _seq(v, 0, lens_k[i]),
                   causal=causal, causal_offset=lens_k[i] - lens_q[i], rows=(0, 256), block_keys=ps["block_keys"], name=name + " vs oracle", split=True)
  det, det_lse = hip.varlen_forward(q, k, v, cu_q, cu_k, max(lens_q), max(lens_k), causal, d ** -0.5, flags=hip.FLAG_DETERMINISTIC)
  assert torch.equal(det, ref) and torch.equal(det_lse, ref_lse)
  env, env_lse = hip.varlen_forward(q, k, v, cu_q, cu_k, max(lens_q), max(lens_k), causal, d ** -0.5, num_splits=1)
  assert torch.equal(env, ref) and torch.equal(env_lse, ref_lse)


@pytest.mark.parametrize("hq, hkv, d, dtype, nq", [(32, 8, 512, torch.bfloat16, 1), (8, 8, 320, torch.float16, 1), (4, 1, 1024, torch.bfloat16, 1), (8, 8, 128, torch.bfloat16, 37)])
def test_decode_batches_with_the_non_temporal_kv_fetch_are_bit_identical(hip, hq, hkv, d, dtype, nq):
  """The NT build of the packed kernel (K / V pieces carry the non-temporal hint: the launch side takes it for decode batches whose K + V have one reader per byte and
  do not fit the caches) computes the same bits as the plain build — the hint changes where a line lives, not what it holds.  Forced either way here (the batch is small)."""
  lens_k = [700, 0, 64, 1300, 129, 2048, 1, 333]
  lens_q = [nq, nq, 0, nq, nq, nq, 1, nq]
  q, k, v = _make(lens_q, lens_k, hq, hkv, d, dtype, seed=hq + d + 1)
  cu_q, cu_k = _cu(lens_q), _cu(lens_k)
  pa, pb = {}, {}
  out_a, lse_a = hip.varlen_forward(q, k, v, cu_q, cu_k, nq, max(lens_k), True, d ** -0.5, flags=hip.FLAG_KV_STREAM, plan_out=pa)
  out_b, lse_b = hip.varlen_forward(q, k, v, cu_q, cu_k, nq, max(lens_k), True, d ** -0.5, flags=hip.FLAG_NO_KV_STREAM, plan_out=pb)
  assert ", NT>" in pa["kernel"] and ", NT>" not in pb["kernel"], (pa, pb)
  assert torch.equal(out_a, out_b) and torch.equal(lse_a, lse_b)
  _check_packed(hip, q, k, v, lens_q, lens_k, True, out_a, lse_a, oracle=False, dense_bits=False, sdpa=d <= 512, name=f"NT packed decode Hq{hq}/Hkv{hkv} D{d}")


@pytest.mark.parametrize("num_splits", [1, 0])
def test_static_capacity_kv_cache_with_device_side_lengths(hip, num_splits):
  """seqused_k (the op-level extension): a KV cache of fixed capacity per sequence, viewed as packed rows, whose valid lengths live on the device — equal, bit for
  bit, to the call on the tightly packed valid rows; ONE captured graph follows lengths rewritten in place; decode under GQA runs packed (one workgroup per
  (sequence, KV head)).  Rows of the cache past a sequence's length are NaN: nothing may read them."""
  b, cap, hq, hkv, d = 6, 1536, 16, 4, 512
  g = torch.Generator(device="cuda").manual_seed(11)
  cache_k = torch.randn((b, cap, hkv, d), dtype=torch.bfloat16, device="cuda", generator=g)
  cache_v = torch.randn((b, cap, hkv, d), dtype=torch.bfloat16, device="cuda", generator=g)
  q = torch.randn((b, hq, d), dtype=torch.bfloat16, device="cuda", generator=g)  # one token per sequence
  cu_q = _cu([1] * b)
  cu_k = torch.arange(0, (b + 1) * cap, cap, dtype=torch.int32, device="cuda")
  used = torch.zeros(b, dtype=torch.int32, device="cuda")

  def poisoned(lens):
    k, v = cache_k.clone(), cache_v.clone()
    for i, n in enumerate(lens):
      k[i, n:], v[i, n:] = float("nan"), float("nan")
    return k.view(b * cap, hkv, d), v.view(b * cap, hkv, d)

  graph = out = lse = None
  kk, vv = poisoned([cap] * b)
  for lens in ([1536, 1, 700, 64, 0, 1000], [5, 1536, 129, 640, 33, 1]):
    kp, vp = poisoned(lens)
    kk.copy_(kp), vv.copy_(vp)
    used.copy_(torch.tensor(lens, dtype=torch.int32))
    if graph is None:
      plan = {}
      hip.varlen_forward(q, kk, vv, cu_q, cu_k, 1, cap, True, d ** -0.5, seqused_k=used, plan_out=plan, num_splits=num_splits)  # (warm-up outside the capture)
      # (24 workgroups for 256 CUs: left to itself the launch splits every sequence's KV range — by its length on the device — and the graph holds two kernels)
      assert plan["workgroups"] == b * hkv * plan["splits"] and "packed into rows" in plan["kernel"] and (plan["splits"] == 1) == (num_splits == 1), plan
      graph = torch.cuda.CUDAGraph()
      with torch.cuda.graph(graph):
        out, lse = hip.varlen_forward(q, kk, vv, cu_q, cu_k, 1, cap, True, d ** -0.5, seqused_k=used, num_splits=num_splits)
    graph.replay()
    torch.cuda.synchronize()
    tight_k = torch.cat([cache_k[i, :n] for i, n in enumerate(lens)]).contiguous()
    tight_v = torch.cat([cache_v[i, :n] for i, n in enumerate(lens)]).contiguous()
    ref, ref_lse = hip.varlen_forward(q, tight_k, tight_v, cu_q, _cu(lens), 1, max(lens), True, d ** -0.5, num_splits=1)
    if num_splits == 1:
      assert torch.equal(out, ref) and torch.equal(lse, ref_lse), lens
    else:
      _same_to_merge_rounding(out, lse, ref, ref_lse, f"cache lens {lens}")
    _check_packed(hip, q, tight_k, tight_v, [1] * b, lens, True, out, lse, oracle=False, dense_bits=False, name=f"cache lens {lens}")
  # the public entry point keeps the reference's refusal of this option
  from ffpa_attn_amd import ffpa_attn_varlen_func

  with pytest.raises(NotImplementedError, match="unsupported options: seqused_k"):
    ffpa_attn_varlen_func(q, kk, vv, cu_q, cu_k, 1, cap, seqused_k=used)
