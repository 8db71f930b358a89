"""CPU-side behaviour of the public API: BASELINE config 1 (SDPA plumbing), monkey-patching,
error contracts, and "large-D CPU tensors raise instead of silently computing somewhere else"."""

import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN
from ffpa_attn_amd import ffpa_attn_func


def _from_bits(a):
  return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)


def test_config1_bit_equal_to_reference_output():
  """B1 H4 N1024 D64 bf16 on CPU: the committed output of the REFERENCE's ffpa_attn_func."""
  z = np.load(os.path.join(GOLDEN, "cfg1_cpu.npz"))
  q, k, v = (_from_bits(z[n]) for n in "qkv")
  torch.manual_seed(0)
  q0 = torch.randn(1, 4, 1024, 64, dtype=torch.bfloat16)
  assert torch.equal(q, q0)  # fixture inputs are exactly "seed 0, randn, q then k then v"
  out = ffpa_attn_func(q, k, v)
  assert torch.equal(out, _from_bits(z["o_reference"]))
  assert torch.equal(out, F.scaled_dot_product_attention(q, k, v))


def test_monkey_patch_does_not_recurse(monkeypatch):
  """tests/test_monkey_patch.py:65-69 in the reference: fallbacks must call the native op."""
  monkeypatch.setattr(F, "scaled_dot_product_attention", ffpa_attn_func)
  q = torch.randn(1, 2, 64, 64, dtype=torch.bfloat16)
  out = F.scaled_dot_product_attention(q, q, q, is_causal=True)
  ref = torch._C._nn.scaled_dot_product_attention(q, q, q, is_causal=True)
  assert torch.equal(out, ref)


def test_large_d_cpu_raises_not_implemented():
  q = torch.randn(1, 2, 512, 320, dtype=torch.bfloat16)
  with pytest.raises(NotImplementedError, match="_fwd_hip"):
    ffpa_attn_func(q, q, q)


def test_error_contracts():
  q = torch.randn(1, 8, 512, 320, dtype=torch.bfloat16)
  k = torch.randn(1, 2, 512, 320, dtype=torch.bfloat16)
  with pytest.raises(TypeError, match="unexpected keyword"):
    ffpa_attn_func(q, q, q, nonsense=True)
  with pytest.raises(ValueError, match="enable_gqa=False"):
    ffpa_attn_func(q, k, k)
  with pytest.raises(ValueError, match="seqlen"):
    ffpa_attn_func(q, q, q[:, :, :500])
  with pytest.raises(ValueError, match="num_heads"):
    ffpa_attn_func(q[:, :6], k[:, :2].repeat(1, 2, 1, 1), k[:, :2].repeat(1, 2, 1, 1), enable_gqa=True)
  with pytest.raises(ValueError, match="is_causal"):
    ffpa_attn_func(q.repeat(1, 1, 2, 1), q, q, is_causal=True)
  with pytest.raises(TypeError, match="fp16/bf16"):
    ffpa_attn_func(q.float(), q.float(), q.float())
  with pytest.raises(RuntimeError, match="attn_mask should not be set"):
    ffpa_attn_func(q, q, q, attn_mask=torch.ones(512, 512, dtype=torch.bool), is_causal=True)
  with pytest.raises((ValueError, TypeError)):
    ffpa_attn_func(q, q, q, backend="nope")


def test_mask_normalisation_shapes():
  from ffpa_attn_amd.functional import FFPAAttnMeta

  q = torch.empty(2, 4, 600, 320, dtype=torch.bfloat16)
  k = torch.empty(2, 4, 700, 320, dtype=torch.bfloat16)
  m = FFPAAttnMeta.from_kwargs()
  # boolean masks keep their bytes (the kernel reads them: FFPA_BIAS_BOOL8) — no 0 / -inf temporary
  mask = torch.ones(600, 700, dtype=torch.bool)
  b = m.normalize_attn_mask(q, k, mask)
  assert b.shape == (1, 1, 600, 700) and b.dtype == torch.bool and b.data_ptr() == mask.data_ptr()
  b = m.normalize_attn_mask(q, k, torch.zeros(2, 1, 700, dtype=torch.bool))
  assert b.shape == (2, 1, 1, 700) and not b.any()
  b = m.normalize_attn_mask(q, k, torch.zeros(2, 4, 1, 700))
  assert b.dtype == torch.float32 and b.shape == (2, 4, 1, 700)


def test_mask_kv_bounds_per_32_row_block():
  """Host side of the mask-derived tile clipping: [first, end) of the keys visible to ANY row of each 32-row block."""
  from ffpa_attn_amd.hip import mask_kv_bounds
  Nq, Nkv = 100, 300
  m = torch.ones(Nq, Nkv, dtype=torch.bool).tril(diagonal=50)
  m[40:70] = False
  bias = torch.zeros(1, 1, Nq, Nkv).masked_fill(~m, float("-inf"))
  assert mask_kv_bounds(bias, Nq, Nkv)[0, 0, :, :2].tolist() == [[0, 82], [0, 90], [0, 146], [0, 150]]
  assert torch.equal(mask_kv_bounds(m.view(1, 1, Nq, Nkv), Nq, Nkv), mask_kv_bounds(bias, Nq, Nkv))  # bool masks: False = hidden
  # neutral interior: keys every row of the block sees unmasked.  block 0 = rows 0..31 (tril +50): keys 0..50; block 1 holds the
  # all-hidden rows 40..63 -> nothing; block 2 (rows 64..95, rows 64..69 hidden) -> nothing; block 3 = rows 96..99: keys 0..146
  assert mask_kv_bounds(bias, Nq, Nkv)[0, 0, :, 2:].tolist() == [[0, 51], [0, 0], [0, 0], [0, 147]]
  win = torch.ones(64, 512, dtype=torch.bool).tril(diagonal=200).triu(diagonal=150)       # sliding window
  b = mask_kv_bounds(torch.zeros(2, 1, 64, 512).masked_fill(~win, float("-inf")), 64, 512)
  assert b.shape == (2, 1, 2, 4) and b.dtype == torch.int32
  assert b[1, 0, :, :2].tolist() == [[150, 232], [182, 264]]
  assert b[1, 0, :, 2:].tolist() == [[181, 201], [213, 233]]                               # keys r+150..r+200 for all of r0..r0+31
  none = mask_kv_bounds(torch.full((1, 1, 32, 64), float("-inf")), 32, 64)                # nothing visible: empty ranges
  assert none[0, 0].tolist() == [[64, 0, 0, 0]]
  pad = mask_kv_bounds(torch.zeros(1, 1, 1, 100).index_fill_(3, torch.arange(60, 100), float("-inf")), 40, 100)   # key padding
  assert pad[0, 0].tolist() == [[0, 60, 0, 60], [0, 60, 0, 60]]
  alibi = mask_kv_bounds(-torch.arange(64.0).view(1, 1, 1, 64).expand(1, 1, 32, 64) * 0.5, 32, 64)  # a real bias: only key 0 adds 0
  assert alibi[0, 0].tolist() == [[0, 64, 0, 1]]


def test_bench_takes_the_kernel_name_from_the_library_not_from_a_copy_of_the_dispatch_rule():
  """bench.py's roofline.kernel is what ffpa_attn_fwd_kernel (C-ABI) reports for the workload's call: no hand-kept copy of
  csrc/ffpa_fwd_inst.hip's dispatch lives in the bench."""
  import os
  import sys

  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  sys.path.insert(0, root)
  import bench

  assert not hasattr(bench, "dominant_kernel")
  src = open(os.path.join(root, "bench.py")).read()
  assert 'plan.get("kernel")' in src and "ffpa_fwd_m16_kernel" not in src and "ffpa_fwd_split_d_kernel" not in src


def test_tensor_version_probe_handles_inference_tensors():
  """hip._tensor_version: torch's in-place counter, None for tensors that do not track one (reading `_version` of an inference
  tensor raises) — the mask-range cache never keys on those."""
  import torch
  from ffpa_attn_amd import hip

  t = torch.ones(4, dtype=torch.bool)
  v0 = hip._tensor_version(t)
  t.fill_(False)
  assert hip._tensor_version(t) == v0 + 1
  with torch.inference_mode():
    m = torch.ones(4, dtype=torch.bool)
  assert hip._tensor_version(m) is None
  assert not hip._mask_bounds_cache_enabled() or True  # (env-dependent: only the call must not raise)


def test_package_version_matches_the_library():
  import ffpa_attn_amd
  from ffpa_attn_amd import hip

  lib = hip.load_library()
  assert lib.ffpa_attn_version().decode().split()[1] == ffpa_attn_amd.__version__


def test_fake_tensors_take_the_registered_op_not_the_launch_wrapper():
  """The inference short cut of `_ffpa_apply` (no autograd node, straight to the launch wrapper) is for real tensors only: fake tensors — export,
  make_fx, shape propagation — must reach `ffpa_attn::_fwd_hip`'s registered fake, on a box without a GPU or the library too."""
  from torch._subclasses.fake_tensor import FakeTensor, FakeTensorMode

  from ffpa_attn_amd import ffpa_attn_func

  with FakeTensorMode():
    q = torch.empty(1, 4, 1024, 512, dtype=torch.bfloat16, device="cuda")
    k = torch.empty(1, 2, 2048, 512, dtype=torch.bfloat16, device="cuda")
    v = torch.empty(1, 2, 2048, 512, dtype=torch.bfloat16, device="cuda")
    o = ffpa_attn_func(q, k, v, is_causal=True, enable_gqa=True)
    mask = torch.empty(1, 1, 1024, 2048, dtype=torch.bool, device="cuda")
    om = ffpa_attn_func(q, k, v, attn_mask=mask, enable_gqa=True)
  for t in (o, om):
    assert isinstance(t, FakeTensor) and t.shape == (1, 4, 1024, 512) and t.dtype == torch.bfloat16 and t.device.type == "cuda"


def test_inference_short_cut_is_for_plain_undecorated_calls_only():
  """`_plain_call` (the gate of `_ffpa_apply`'s inference short cut): an active TorchDispatchMode (FlopCounterMode, profiler op records), a functorch
  transform (vmap's BatchedTensor has type torch.Tensor) or a subclass / fake MASK must reach the registered op, not the ctypes launch wrapper."""
  from torch.utils.flop_counter import FlopCounterMode

  from ffpa_attn_amd.interface import _plain_call

  q = torch.zeros(1, 1, 8, 8)
  assert _plain_call(q, q, q, None) and _plain_call(q, q, q, torch.zeros(8, 8, dtype=torch.bool))
  with FlopCounterMode(display=False):
    assert not _plain_call(q, q, q, None)
  seen = []

  def f(x):
    seen.append(_plain_call(x, x, x, None))
    return x

  torch.vmap(f)(torch.zeros(2, 1, 1, 8, 8))
  assert seen == [False]

  class Sub(torch.Tensor):
    pass

  assert not _plain_call(q, q, q, torch.zeros(8, 8).as_subclass(Sub))
  assert not _plain_call(q.as_subclass(Sub), q, q, None)
  assert _plain_call(q, q, q, None)  # (nothing leaked from the modes above)
