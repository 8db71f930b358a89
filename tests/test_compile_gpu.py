"""torch.compile / torch.library coverage of the HIP op (`pytest -m gpu`), after the reference's tests/test_ffpa_compile.py:52-89 (every
forward backend and the forward / backward pairs under torch.compile): the public function compiles with the graph break the reference
has at the same place (`_ffpa_apply`, functional.py:1195-1216), the registered op traces through its fake implementation without a graph
break, and `torch.library.opcheck` accepts its schema / fake-tensor / dispatch registration."""
import pytest
import torch
import torch.nn.functional as F

from test_fwd_gpu import _rand, hip  # noqa: F401  (fixture + helpers)

pytestmark = pytest.mark.gpu


def _op_args(q, k, v, bias=None, causal=0, dropout=0.0):
  return (q, k, v, q.new_empty((0,)) if bias is None else bias, 0, 1, causal, q.size(-1) ** -0.5, dropout, 0, 0)


def test_opcheck_of_the_registered_op(hip):
  """torch.library.opcheck: schema (no undeclared mutation / aliasing), fake-tensor implementation (shapes, dtypes, strides of both
  outputs), dispatch registration — for an unmasked, a masked-GQA and a decode-shaped call."""
  q, k, v = _rand((1, 4, 520, 320), seed=1), _rand((1, 2, 600, 320), seed=2), _rand((1, 2, 600, 320), seed=3)
  mask = torch.rand(1, 1, 520, 600, device="cuda") > 0.2
  mask[..., 0] = True
  qd = _rand((2, 4, 1, 512), seed=4)
  kd, vd = _rand((2, 4, 2048, 512), seed=5), _rand((2, 4, 2048, 512), seed=6)
  for args in (_op_args(q, k, v, causal=1), _op_args(q, k, v, bias=mask), _op_args(qd, kd, vd)):
    res = torch.library.opcheck(torch.ops.ffpa_attn._fwd_hip, args, test_utils=("test_schema", "test_faketensor"), raise_exception=True)
    assert all(v_ == "SUCCESS" for v_ in res.values()), res


def test_the_raw_op_compiles_without_a_graph_break(hip):
  q, k, v = _rand((1, 4, 520, 512), seed=11), _rand((1, 2, 777, 512), seed=12), _rand((1, 2, 777, 512), seed=13)

  def f(q, k, v):
    o, lse = torch.ops.ffpa_attn._fwd_hip(*_op_args(q, k, v, causal=1))
    return o * 2.0, lse + 1.0

  o_e, l_e = f(q, k, v)
  o_c, l_c = torch.compile(f, fullgraph=True)(q, k, v)
  assert torch.equal(o_e, o_c) and torch.equal(l_e, l_c)


@pytest.mark.parametrize("case", ["mask_gqa", "causal_cross_gqa", "key_bias", "d1024"])
def test_compiled_forward_cases(hip, case):
  """Masked / GQA / biased calls of the public function under torch.compile: the same bits as eager."""
  from ffpa_attn_amd import ffpa_attn_func

  D = 1024 if case == "d1024" else 512
  q, k, v = _rand((2, 8, 640, D), seed=21), _rand((2, 2, 900, D), seed=22), _rand((2, 2, 900, D), seed=23)
  kw = dict(enable_gqa=True)
  if case == "mask_gqa":
    m = torch.rand(2, 1, 640, 900, device="cuda") > 0.3
    m[..., 0] = True
    kw["attn_mask"] = m
  elif case == "causal_cross_gqa":
    kw["is_causal"] = True
  elif case == "key_bias":
    kw["attn_mask"] = _rand((1, 1, 1, 900), seed=24) * 0.5
  eager = ffpa_attn_func(q, k, v, **kw)
  compiled = torch.compile(lambda a, b, c: ffpa_attn_func(a, b, c, **kw))(q, k, v)
  assert torch.equal(eager, compiled), case
  if "attn_mask" in kw and kw["attn_mask"].dtype != torch.bool:
    return
  ref = F.scaled_dot_product_attention(q, k, v, attn_mask=kw.get("attn_mask"), is_causal=False, enable_gqa=True) if case == "mask_gqa" else None
  if ref is not None:
    assert (eager.float() - ref.float()).abs().max().item() <= 2e-2


@pytest.mark.parametrize("causal", [False, True])
def test_compiled_forward_backward_pair(hip, causal):
  """Forward + backward under torch.compile (the reference's fwd / bwd pairs, tests/test_ffpa_compile.py:72-89): gradients equal the eager
  run's bit for bit (the same kernels run on both sides of the graph break) and agree with autograd through fp32 math."""
  from ffpa_attn_amd import ffpa_attn_func

  D = 512
  base = [_rand((1, 4, 600, D), seed=31), _rand((1, 2, 600, D), seed=32), _rand((1, 2, 600, D), seed=33)]

  def loss_fn(q, k, v):
    return (ffpa_attn_func(q, k, v, is_causal=causal, enable_gqa=True).float() ** 2).sum()

  grads = {}
  for name, fn in (("eager", loss_fn), ("compiled", torch.compile(loss_fn))):
    qkv = [t.clone().requires_grad_() for t in base]
    fn(*qkv).backward()
    grads[name] = [t.grad for t in qkv]
  for a, b in zip(grads["eager"], grads["compiled"]):
    assert torch.equal(a, b)
  qkv = [t.clone().float().requires_grad_() for t in base]
  kf, vf = qkv[1].repeat_interleave(2, 1), qkv[2].repeat_interleave(2, 1)
  s = (qkv[0] @ kf.transpose(-1, -2)) * D ** -0.5
  if causal:
    r, c = torch.arange(600, device="cuda")[:, None], torch.arange(600, device="cuda")[None, :]
    s = s.masked_fill(c > r, float("-inf"))
  ((torch.softmax(s, -1) @ vf) ** 2).sum().backward()
  for g, t in zip(grads["compiled"], qkv):
    rel = (g.float() - t.grad).abs().max().item() / max(t.grad.abs().max().item(), 1e-6)
    assert rel <= 3e-2, rel
