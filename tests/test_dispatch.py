"""Host-side dispatch / validation parity with the reference, on meta tensors (no GPU).

tests/golden/dispatch_golden.json was produced by running the REFERENCE's
FFPAAttnMeta.{from_kwargs,fallback,normalize} (src/ffpa_attn/functional.py:611-943) on the same
cases in the authoring container; here this repo's FFPAAttnMeta must take the same decision:
fall back to SDPA, raise the same exception class with the same message, or go to the fused
kernel with the same resolved softmax scale.
"""

import json
import os

import pytest
import torch

from conftest import GOLDEN
from ffpa_attn_amd.functional import FFPAAttnMeta

_DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32, "bool": torch.bool}
CASES = json.load(open(os.path.join(GOLDEN, "dispatch_golden.json")))


def _decide(c):
  dt = _DT[c["dtype"]]
  q = torch.empty(c["q"], dtype=dt, device="meta")
  k = torch.empty(c["k"], dtype=dt, device="meta")
  v = torch.empty(c["v"], dtype=dt, device="meta")
  mask = None
  if c["mask"] is not None:
    mask = torch.empty(c["mask"]["shape"], dtype=_DT[c["mask"]["dtype"]], device="meta")
  kw = dict(c["kwargs"])
  dropout_p = kw.pop("dropout_p", 0.0)
  is_causal = kw.pop("is_causal", False)
  scale = kw.pop("scale", None)
  enable_gqa = kw.pop("enable_gqa", False)
  meta = FFPAAttnMeta.from_kwargs(**kw)
  if meta.fallback(q, k, mask, dropout_p):
    return {"fallback": True}
  meta, *_ = meta.normalize(q, k, v, mask, dropout_p, is_causal, scale, enable_gqa)
  return {"ffpa": {"scale": meta.attn_meta.scale}}


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_same_decision_as_reference(c):
  want = c["expect"]
  if "raises" in want:
    exc = {"ValueError": ValueError, "TypeError": TypeError, "RuntimeError": RuntimeError,
           "NotImplementedError": NotImplementedError, "AssertionError": AssertionError}[want["raises"]["type"]]
    with pytest.raises(exc) as ei:
      _decide(c)
    ref_msg = want["raises"]["message"]
    got_msg = str(ei.value)
    if c["name"] == "bad_backend_str":  # this build also accepts 'hip', so the list in the message is longer
      assert "must be 'cuda', 'triton', 'cutedsl', or 'sdpa'" in got_msg and "'nope'" in got_msg
    else:
      assert got_msg == ref_msg
  else:
    got = _decide(c)
    assert set(got) == set(want), (got, want)
    if "ffpa" in want:
      assert got["ffpa"]["scale"] == pytest.approx(want["ffpa"]["scale"], rel=1e-12)


def test_backend_objects_and_aliases():
  from ffpa_attn_amd import CUDABackend, HIPBackend, SDPABackend, TritonBackend

  m = FFPAAttnMeta.from_kwargs()
  assert m.forward_meta.name == "hip" and m.forward_meta.forward
  for name in ("cuda", "triton", "hip"):
    m = FFPAAttnMeta.from_kwargs(forward_backend=name)
    assert isinstance(m.forward_meta, HIPBackend) and m.forward_meta.name == name
  m = FFPAAttnMeta.from_kwargs(forward_backend=CUDABackend(forward=True), backward_backend=SDPABackend(backward=True))
  assert m.forward_meta.name == "cuda" and m.backward_meta.name == "sdpa"
  with pytest.raises(ValueError):
    HIPBackend(acc="f16")
  with pytest.raises(NotImplementedError):
    CUDABackend(forward=True, enable_fp8=True)
  assert TritonBackend(forward=True).backward is False
  # the reference's CUDA backend is forward-only (functional.py:266-268): its default construction asserts
  with pytest.raises(AssertionError, match="cuda backend does not support backward"):
    CUDABackend()
  with pytest.raises(AssertionError, match="cuda backend does not support backward"):
    CUDABackend(backward=True)
  assert CUDABackend(forward=True).backward is False and CUDABackend(backward=False).forward is True


def test_small_d_opt_in_env(monkeypatch):
  q = torch.empty((1, 8, 1024, 128), dtype=torch.bfloat16, device="meta")
  assert FFPAAttnMeta.from_kwargs().fallback(q, q, None, 0.0)
  monkeypatch.setenv("FFPA_HIP_ALLOW_SMALL_D", "1")
  assert not FFPAAttnMeta.from_kwargs().fallback(q, q, None, 0.0)
  q32 = torch.empty((1, 8, 1024, 32), dtype=torch.bfloat16, device="meta")
  assert FFPAAttnMeta.from_kwargs().fallback(q32, q32, None, 0.0)  # below 64 never reaches the kernel
