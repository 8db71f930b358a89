"""The reference's own forward test matrix, re-run against this build (GPU).

Every case below is a (shape, feature) point that xlite-dev/ffpa-attn's tests/test_ffpa_fwd.py exercises —
its correctness / dispatch / boundary / cross / GQA / causal / mask / decode grids (:32-46, :291-336,
:897-1018, :1084-1094, :1110-1121, :1160-1174, :1226-1250, :1307-1320) — checked the way it checks them:
`assert_close` to PyTorch SDPA on seeded randn at atol = rtol = 2e-2 (bf16) / 1e-2 (fp16).  Each point runs
through two routes:

  api     ffpa_attn_func, i.e. with the reference's fallback rules (small D, 8 <= Nq < 512, Nkv < 512 go to SDPA)
  kernel  the gfx950 kernel through the C-ABI regardless of those rules (what FFPA_HIP_ALLOW_SMALL_D / a direct
          op call reach), so that the whole matrix also lands on the hand-written path

Only the shapes are taken from the reference (they are data); the checks are written against this repo's API.
"""
import itertools

import pytest
import torch

pytestmark = pytest.mark.gpu

DTYPES = [torch.float16, torch.bfloat16]
IDS = ["fp16", "bf16"]


@pytest.fixture(scope="module")
def hip():
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  from ffpa_attn_amd import hip as h
  h.load_library()  # fail loudly if the extension is missing
  return h


def _tol(dtype):
  return dict(atol=2e-2, rtol=2e-2) if dtype == torch.bfloat16 else dict(atol=1e-2, rtol=1e-2)


def _qkv(B, Hq, Hkv, Nq, Nkv, D, dtype, seed=0):
  g = torch.Generator(device="cuda").manual_seed(seed)
  q = torch.randn(B, Hq, Nq, D, dtype=dtype, device="cuda", generator=g)
  k = torch.randn(B, Hkv, Nkv, D, dtype=dtype, device="cuda", generator=g)
  v = torch.randn(B, Hkv, Nkv, D, dtype=dtype, device="cuda", generator=g)
  return q, k, v


def _tail_causal_mask(Nq, Nkv):
  rows = torch.arange(Nq, device="cuda")[:, None]
  cols = torch.arange(Nkv, device="cuda")[None, :]
  return cols <= rows + (Nkv - Nq)


def _sdpa(q, k, v, *, mask=None, causal=False):
  """SDPA reference.  The reference's causal convention is tail-aligned: expressed as an explicit mask whenever
  Nq != Nkv (PyTorch's is_causal is top-left)."""
  gqa = q.size(1) != k.size(1)
  if causal and q.size(2) != k.size(2):
    mask, causal = _tail_causal_mask(q.size(2), k.size(2)), False
  return torch._C._nn.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, is_causal=causal,
                                                   scale=None, enable_gqa=gqa)


def _run(route, hip, q, k, v, *, mask=None, causal=False):
  gqa = q.size(1) != k.size(1)
  if route == "api":
    from ffpa_attn_amd import ffpa_attn_func
    return ffpa_attn_func(q, k, v, attn_mask=mask, is_causal=causal, enable_gqa=gqa)
  bias = None
  if mask is not None:  # what FFPAAttnMeta.normalize_attn_mask hands the op: additive, 4-D, q.dtype or fp32
    bias = torch.zeros(mask.shape, dtype=q.dtype, device="cuda").masked_fill(~mask, float("-inf")) if mask.dtype == torch.bool else mask
    while bias.dim() < 4:
      bias = bias.unsqueeze(0)
  o, _ = hip.forward(q, k, v, bias, causal, q.size(-1) ** -0.5)
  return o


def _falls_back(q, k):
  """The reference's fallback predicate (functional.py:676-724) for the default backend and environment."""
  D, Nq, Nkv = q.size(-1), q.size(2), k.size(2)
  return D <= 256 or D > 1024 or 8 <= Nq < 512 or Nkv < 512


def _check(route, hip, q, k, v, **kw):
  out = _run(route, hip, q, k, v, **kw)
  if route == "api" and _falls_back(q, k):
    # fallback shapes ARE raw SDPA in the reference, including its top-left is_causal for Nq != Nkv — which is
    # what the reference's own tests expect there (tests/test_ffpa_fwd.py:1273-1287)
    ref = torch._C._nn.scaled_dot_product_attention(q, k, v, attn_mask=kw.get("mask"), dropout_p=0.0, is_causal=kw.get("causal", False),
                                                    scale=None, enable_gqa=q.size(1) != k.size(1))
    assert torch.equal(out, ref)
    return
  ref = _sdpa(q, k, v, **kw)
  assert out.dtype == q.dtype and out.shape == ref.shape
  assert torch.isfinite(out).all()
  torch.testing.assert_close(out, ref, **_tol(q.dtype))


ROUTES = pytest.mark.parametrize("route", ["api", "kernel"])
BY_DTYPE = pytest.mark.parametrize("dtype", DTYPES, ids=IDS)


# --------------------------------------------------------------------------- self-attention grids
@ROUTES
@BY_DTYPE
@pytest.mark.parametrize("B,H,N,D", [(1, 8, 1024, 64), (1, 8, 1024, 128), (1, 16, 1024, 320), (1, 16, 1024, 512),
                                     (1, 32, 1024, 640), (1, 32, 4096, 128), (1, 48, 4096, 320)])
def test_matches_sdpa(hip, route, dtype, B, H, N, D):
  _check(route, hip, *_qkv(B, H, H, N, N, D, dtype))


@ROUTES
@BY_DTYPE
@pytest.mark.parametrize("H,D", list(itertools.product([8, 16, 32, 48], [64, 128, 320, 512, 640])))
def test_every_head_count_and_head_dim_launches(hip, route, dtype, H, D):
  q, k, v = _qkv(1, H, H, 1024, 1024, D, dtype)
  out = _run(route, hip, q, k, v)
  assert out.shape == q.shape and out.dtype == dtype and torch.isfinite(out).all()


@ROUTES
@BY_DTYPE
@pytest.mark.parametrize("N,D", list(itertools.product([1, 17, 33, 63, 65, 100, 127, 129, 200, 1000, 2047, 4095, 5000], [128, 256])))
def test_boundary_sequence_lengths(hip, route, dtype, N, D):
  _check(route, hip, *_qkv(1, 4, 4, N, N, D, dtype))


# --------------------------------------------------------------------------- cross attention, GQA
@ROUTES
@BY_DTYPE
@pytest.mark.parametrize("D", [128, 256, 512])
@pytest.mark.parametrize("Nq,Nkv", [(128, 1024), (128, 8192), (1024, 128), (1024, 8192), (8191, 8192), (8192, 8191), (1, 4096)])
def test_cross_attention(hip, route, dtype, Nq, Nkv, D):
  _check(route, hip, *_qkv(1, 4, 4, Nq, Nkv, D, dtype))


@ROUTES
@BY_DTYPE
@pytest.mark.parametrize("D", [64, 128, 256, 512])
@pytest.mark.parametrize("Nq,Nkv", [(128, 128), (1024, 1024), (128, 8192), (1024, 4096)])
@pytest.mark.parametrize("Hq,Hkv", [(8, 1), (16, 2), (32, 4), (32, 8), (16, 16)])
def test_gqa(hip, route, dtype, Hq, Hkv, Nq, Nkv, D):
  _check(route, hip, *_qkv(1, Hq, Hkv, Nq, Nkv, D, dtype))


# --------------------------------------------------------------------------- causal (tail-aligned, the reference's convention)
@ROUTES
@BY_DTYPE
@pytest.mark.parametrize("N,D", [(64, 128), (128, 64), (1024, 128), (1024, 256), (4096, 128), (127, 128), (129, 256),
                                 (512, 320), (1024, 512), (512, 1024)])
def test_causal_self_attention(hip, route, dtype, N, D):
  _check(route, hip, *_qkv(1, 4, 4, N, N, D, dtype), causal=True)


@ROUTES
@BY_DTYPE
@pytest.mark.parametrize("Nq,Nkv,D", [(1, 8192, 128), (128, 1024, 128), (128, 8192, 256), (1024, 4096, 128), (129, 2048, 512),
                                      (128, 4096, 512), (64, 2048, 1024)])
def test_causal_cross_attention(hip, route, dtype, Nq, Nkv, D):
  _check(route, hip, *_qkv(1, 4, 4, Nq, Nkv, D, dtype), causal=True)


@ROUTES
@BY_DTYPE
@pytest.mark.parametrize("Nq,Nkv,D", [(128, 128, 128), (1024, 1024, 128), (128, 4096, 256), (1024, 1024, 512), (128, 4096, 512),
                                      (512, 2048, 1024)])
@pytest.mark.parametrize("Hq,Hkv", [(8, 1), (32, 4), (32, 8)])
def test_causal_gqa(hip, route, dtype, Hq, Hkv, Nq, Nkv, D):
  _check(route, hip, *_qkv(1, Hq, Hkv, Nq, Nkv, D, dtype), causal=True)


# --------------------------------------------------------------------------- masks
@ROUTES
@BY_DTYPE
@pytest.mark.parametrize("D", [320, 512])
@pytest.mark.parametrize("kind", ["bool_2d", "additive_broadcast"])
def test_attn_mask(hip, route, dtype, D, kind):
  q, k, v = _qkv(1, 4, 4, 512, 512, D, dtype)
  if kind == "bool_2d":
    mask = torch.ones(512, 512, dtype=torch.bool, device="cuda")
    mask[:, 3::7] = False
    mask[:, 0] = True
  else:
    mask = torch.randn(1, 1, 1, 512, device="cuda", dtype=dtype, generator=torch.Generator(device="cuda").manual_seed(1)) * 0.25
  _check(route, hip, q, k, v, mask=mask)


@ROUTES
def test_attn_mask_cross_gqa(hip, route):
  q, k, v = _qkv(1, 4, 2, 512, 768, 320, torch.float16)
  mask = torch.randn(1, 1, 512, 768, device="cuda", dtype=torch.float16, generator=torch.Generator(device="cuda").manual_seed(1)) * 0.125
  _check(route, hip, q, k, v, mask=mask)


# --------------------------------------------------------------------------- decode / short query
@ROUTES
@BY_DTYPE
@pytest.mark.parametrize("Nq,Nkv,D,causal", [(1, 4096, 512, False), (7, 4096, 320, False), (1, 8192, 512, False), (7, 8192, 512, True),
                                             (15, 8192, 512, True), (32, 8192, 512, False), (512, 8192, 512, False)])
def test_decode_and_short_query(hip, route, dtype, Nq, Nkv, D, causal):
  _check(route, hip, *_qkv(1, 4, 4, Nq, Nkv, D, dtype), causal=causal)


@ROUTES
@BY_DTYPE
@pytest.mark.parametrize("Nq,D", [(1, 512), (5, 320), (7, 320)])
def test_decode_gqa(hip, route, dtype, Nq, D):
  _check(route, hip, *_qkv(2, 32, 8, Nq, 4096, D, dtype))


# --------------------------------------------------------------------------- backward through the saved O / LSE (backward_backend="sdpa")
def _grads(fn, q, k, v, go):
  q, k, v = (t.detach().clone().requires_grad_() for t in (q, k, v))
  return torch.autograd.grad(fn(q, k, v), [q, k, v], go)


@BY_DTYPE
@pytest.mark.parametrize("Hq,Hkv,N,D,causal", [(8, 8, 4096, 320, False), (8, 8, 4096, 512, False), (16, 16, 8192, 320, False),
                                               (4, 4, 4096, 320, True), (4, 4, 8192, 512, True),
                                               (32, 4, 8192, 320, False), (8, 1, 8192, 320, False), (32, 8, 4096, 512, True)])
def test_backward_matches_sdpa_autograd(hip, dtype, Hq, Hkv, N, D, causal):
  """dQ / dK / dV of ffpa_attn_func (HIP forward, PyTorch efficient-attention backward on its O and LSE) vs autograd
  through SDPA on the same inputs — the reference's backward grids (tests/test_ffpa_bwd.py:896-903,935-941,1070-1072)."""
  from ffpa_attn_amd import ffpa_attn_func
  q, k, v = _qkv(1, Hq, Hkv, N, N, D, dtype)
  go = torch.randn(q.shape, dtype=dtype, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
  gqa = Hq != Hkv
  got = _grads(lambda a, b, c: ffpa_attn_func(a, b, c, is_causal=causal, enable_gqa=gqa), q, k, v, go)
  want = _grads(lambda a, b, c: torch.nn.functional.scaled_dot_product_attention(a, b, c, is_causal=causal, enable_gqa=gqa), q, k, v, go)
  for name, a, b in zip(("dq", "dk", "dv"), got, want):
    assert a.dtype == dtype and a.shape == b.shape
    torch.testing.assert_close(a, b, **_tol(dtype), msg=lambda m: f"{name}: {m}")


# --------------------------------------------------------------------------- the api route really is the kernel where the rules say so
def test_api_route_reaches_native_sdpa_only_where_the_reference_falls_back(hip, monkeypatch):
  from ffpa_attn_amd import ffpa_attn_func
  calls = []
  real = torch._C._nn.scaled_dot_product_attention
  monkeypatch.setattr(torch._C._nn, "scaled_dot_product_attention", lambda *a, **kw: calls.append(1) or real(*a, **kw))
  for D, n_fallback in ((512, 0), (320, 0), (1024, 0), (128, 1), (256, 1)):
    calls.clear()
    q, k, v = _qkv(1, 8, 8, 1024, 1024, D, torch.bfloat16)
    ffpa_attn_func(q, k, v)
    assert len(calls) == n_fallback, (D, calls)
  monkeypatch.setenv("FFPA_HIP_ALLOW_SMALL_D", "1")
  calls.clear()
  q, k, v = _qkv(1, 8, 8, 1024, 1024, 128, torch.bfloat16)
  ffpa_attn_func(q, k, v)
  assert not calls
