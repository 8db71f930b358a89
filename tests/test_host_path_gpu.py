"""The host side of a launch (`pytest -m gpu`): what a call costs before the kernel starts and what it keeps between calls.

Round 5: an inference call (nothing to differentiate) goes straight from `ffpa_attn_func` to the launch wrapper — no autograd node, no dispatcher
round trip, no LSE tensor; workspace bytes / ticket counts are asked of the library once per shape class; the scratch of short-query (decode)
launches is kept per (device, stream); precomputed mask ranges travel through the Backend object; FFPA_HIP_PREFILL_SPLITS=0 opts out of the
prefill KV-split rules; the split pricing reads the device it runs on.
"""

import os

import pytest
import torch

from test_fwd_gpu import _rand, hip  # noqa: F401  (fixture + helpers)

pytestmark = pytest.mark.gpu


def test_inference_calls_skip_autograd_and_equal_the_training_path(hip):
  from ffpa_attn_amd import ffpa_attn_func

  q, k, v = _rand((1, 4, 640, 512), seed=1), _rand((1, 2, 1024, 512), seed=2), _rand((1, 2, 1024, 512), seed=3)
  o_inf = ffpa_attn_func(q, k, v, is_causal=True, enable_gqa=True)
  assert o_inf.grad_fn is None
  with torch.no_grad():
    qg = q.clone().requires_grad_(True)
    o_ng = ffpa_attn_func(qg, k, v, is_causal=True, enable_gqa=True)  # grad mode off: still the direct path
  assert o_ng.grad_fn is None and torch.equal(o_ng, o_inf)
  qg = q.clone().requires_grad_(True)
  o_tr = ffpa_attn_func(qg, k, v, is_causal=True, enable_gqa=True)
  assert o_tr.grad_fn is not None and torch.equal(o_tr.detach(), o_inf)  # the same launch behind the autograd node
  o_tr.float().square().mean().backward()
  assert qg.grad is not None and torch.isfinite(qg.grad).all()
  # dropout: both paths reserve the generator offsets the same way -> the same keep mask
  torch.manual_seed(5)
  d_inf = ffpa_attn_func(q, k, v, dropout_p=0.2, enable_gqa=True)
  torch.manual_seed(5)
  d_tr = ffpa_attn_func(qg, k, v, dropout_p=0.2, enable_gqa=True)
  assert torch.equal(d_inf, d_tr.detach()) and not torch.equal(d_inf, o_inf)


def test_decode_scratch_is_asked_once_and_kept_per_stream(hip):
  from ffpa_attn_amd import ffpa_attn_func

  q, k, v = _rand((2, 32, 1, 512), seed=11), _rand((2, 8, 8192, 512), seed=12), _rand((2, 8, 8192, 512), seed=13)
  hip._WORKSPACES.clear()
  hip._PLAN_SCRATCH.clear()
  plan = {}
  hip.forward(q, k, v, None, False, 512 ** -0.5, plan_out=plan)
  assert plan["variant"] == 1 and plan["splits"] > 1, plan
  o1 = ffpa_attn_func(q, k, v, enable_gqa=True)
  cur = torch.cuda.current_stream().cuda_stream
  assert len(hip._PLAN_SCRATCH) == 1 and list(hip._WORKSPACES) == [(q.device.index, cur)]
  ws = hip._WORKSPACES[(q.device.index, cur)]
  ptr = ws.data_ptr()
  for _ in range(3):
    o2 = ffpa_attn_func(q, k, v, enable_gqa=True)
  assert hip._WORKSPACES[(q.device.index, cur)].data_ptr() == ptr and len(hip._PLAN_SCRATCH) == 1 and torch.equal(o1, o2)
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):
    o3 = ffpa_attn_func(q, k, v, enable_gqa=True)  # another stream may overlap: its own scratch
  side.synchronize()
  assert len(hip._WORKSPACES) == 2 and hip._WORKSPACES[(q.device.index, side.cuda_stream)].data_ptr() != ptr and torch.equal(o1, o3)
  # a prefill launch that splits the KV axis needs hundreds of MiB of partials: those go back to the caching allocator
  qp, kp, vp = _rand((1, 9, 4096, 512), seed=14), _rand((1, 9, 8192, 512), seed=15), _rand((1, 9, 8192, 512), seed=16)
  hip.forward(qp, kp, vp, None, False, 512 ** -0.5, plan_out=plan)
  assert plan["variant"] == 0 and plan["splits"] > 1 and hip._WORKSPACES[(q.device.index, cur)].data_ptr() == ptr
  # streams come and go: the tables are bounded
  for _ in range(hip._SCRATCH_MAX_STREAMS + 4):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
      ffpa_attn_func(q, k, v, enable_gqa=True)
    s.synchronize()
  assert len(hip._WORKSPACES) <= hip._SCRATCH_MAX_STREAMS


def test_precomputed_mask_ranges_travel_through_the_backend_object(hip):
  from ffpa_attn_amd import CUDABackend, HIPBackend, TritonBackend, ffpa_attn_func

  Nq, Nkv = 1024, 2048
  q, k, v = _rand((1, 4, Nq, 320), seed=21), _rand((1, 2, Nkv, 320), seed=22), _rand((1, 2, Nkv, 320), seed=23)
  mask = torch.ones(Nq, Nkv, dtype=torch.bool, device="cuda").tril(diagonal=Nkv - Nq)
  ranges = hip.mask_kv_bounds(mask.view(1, 1, Nq, Nkv), Nq, Nkv)
  ref = ffpa_attn_func(q, k, v, attn_mask=mask, enable_gqa=True)  # ranges derived by the scan, every call
  calls = []
  real = hip.mask_kv_bounds
  hip.mask_kv_bounds = lambda *a, **kw: calls.append(1) or real(*a, **kw)
  try:
    for be in (HIPBackend(forward=True, kv_bounds=ranges), TritonBackend(forward=True, kv_bounds=ranges), CUDABackend(forward=True, kv_bounds=ranges)):
      assert torch.equal(ffpa_attn_func(q, k, v, attn_mask=mask, enable_gqa=True, forward_backend=be), ref)
      qg = q.clone().requires_grad_(True)  # ... and through the registered op (the path behind the autograd node)
      assert torch.equal(ffpa_attn_func(qg, k, v, attn_mask=mask, enable_gqa=True, forward_backend=be).detach(), ref)
    assert not calls  # no scan ran
    ffpa_attn_func(q, k, v, attn_mask=mask, enable_gqa=True)
    assert calls == [1]
  finally:
    hip.mask_kv_bounds = real
  assert torch.equal(ffpa_attn_func(q, k, v, is_causal=True, enable_gqa=True), ref)  # the explicit tail-aligned mask == the flag


def test_prefill_split_opt_out(hip, monkeypatch):
  q, k, v = _rand((1, 9, 4096, 512), seed=31), _rand((1, 9, 8192, 512), seed=32), _rand((1, 9, 8192, 512), seed=33)
  plan = {}
  o_split, _ = hip.forward(q, k, v, None, False, 512 ** -0.5, plan_out=plan)
  assert plan["splits"] > 1, plan  # a ragged round (288 workgroups on 256 CUs)
  monkeypatch.setenv("FFPA_HIP_PREFILL_SPLITS", "0")
  o_plain, _ = hip.forward(q, k, v, None, False, 512 ** -0.5, plan_out=plan)
  assert plan["splits"] == 1
  o_one, _ = hip.forward(q[:, :1].contiguous(), k[:, :1].contiguous(), v[:, :1].contiguous(), None, False, 512 ** -0.5, num_splits=1)
  assert torch.equal(o_plain[:, :1], o_one)  # without splits a head's bits do not depend on how many heads share the launch
  assert (o_split.float() - o_plain.float()).abs().max().item() <= 4e-3
  qd, kd, vd = _rand((1, 8, 1, 512), seed=34), _rand((1, 8, 8192, 512), seed=35), _rand((1, 8, 8192, 512), seed=36)
  hip.forward(qd, kd, vd, None, False, 512 ** -0.5, plan_out=plan)
  assert plan["variant"] == 1 and plan["splits"] > 1  # short-query launches keep their rule


def test_the_pricing_reads_the_device(hip):
  lib = hip.load_library()
  props = torch.cuda.get_device_properties(0)
  assert lib.ffpa_attn_query(8) == props.multi_processor_count
  mhz, gbps = lib.ffpa_attn_query(9), lib.ffpa_attn_query(10)
  print(f"device pricing inputs: {props.multi_processor_count} CUs, {mhz} MHz, {gbps} GB/s HBM")
  assert 1000 <= mhz <= 4000 and 500 <= gbps <= 20000
  pl = hip.launch_plan(1, 32, 32, 8192, 8192, 512, device=0)
  assert pl["splits"] == 1 and pl["block_rows"] == 128 and pl["kernel"].startswith("ffpa_fwd_m16_kernel<bf16, 512, MK=0")
  pl = hip.launch_plan(2, 32, 8, 8192, 2048, 320, bias_dtype=torch.bool, device=0)
  assert pl["kernel"].startswith("ffpa_fwd_m16w_kernel<bf16, 320, RH=3, MK=2") and pl["block_rows"] == 192 and pl["block_keys"] == 64, pl


def test_decode_step_replays_equal_eager_calls_bit_for_bit(hip):
  """`DecodeStep` (ffpa_attn_amd/decode.py): the step captured into a HIP graph once per (tensors, shapes, stream) and replayed — equal to the plain call
  bit for bit on three shapes (MHA decode, GQA decode with packed heads, a 4-row speculative step with a static boolean key mask), again after the
  inputs' CONTENT changed in place, again after the KV length changed (another key: another capture), and the entries are bounded (LRU)."""
  from ffpa_attn_amd import DecodeStep, ffpa_attn_func

  shapes = [
    dict(q=(1, 32, 1, 512), kv=(1, 32, 4096, 512), gqa=False, mask=False),
    dict(q=(2, 32, 1, 512), kv=(2, 8, 8192, 512), gqa=True, mask=False),
    dict(q=(1, 8, 4, 320), kv=(1, 8, 2048, 320), gqa=False, mask=True),
  ]
  for i, sh in enumerate(shapes):
    step = DecodeStep(enable_gqa=sh["gqa"], max_graphs=2)
    q, k, v = _rand(sh["q"], seed=40 + i), _rand(sh["kv"], seed=50 + i), _rand(sh["kv"], seed=60 + i)
    mask = None
    if sh["mask"]:  # a static-capacity cache whose valid length lives in a device-side mask: its content changes between replays, its address does not
      mask = torch.zeros(1, 1, 1, sh["kv"][2], dtype=torch.bool, device=q.device)
      mask[..., :1500] = True
    want = ffpa_attn_func(q, k, v, attn_mask=mask, enable_gqa=sh["gqa"])
    got = step(q, k, v, mask)
    assert step.captures == 1 and len(step) == 1 and torch.equal(got, want), sh
    for _ in range(3):
      assert torch.equal(step(q, k, v, mask), want)
    assert step.captures == 1
    # new content at the same addresses (the next token's q, a longer valid length): the replay reads it
    q.copy_(_rand(sh["q"], seed=70 + i))
    if mask is not None:
      mask[..., 1500:1777] = True
    want2 = ffpa_attn_func(q, k, v, attn_mask=mask, enable_gqa=sh["gqa"])
    assert not torch.equal(want2, want) and torch.equal(step(q, k, v, mask), want2) and step.captures == 1
    # a KV-length change is another key: captured once, the old entry stays valid
    n2 = sh["kv"][2] - 128
    k2, v2 = k[:, :, :n2], v[:, :, :n2]
    m2 = mask[..., :n2] if mask is not None else None
    want3 = ffpa_attn_func(q, k2, v2, attn_mask=m2, enable_gqa=sh["gqa"])
    assert torch.equal(step(q, k2, v2, m2), want3) and step.captures == 2 and len(step) == 2
    assert torch.equal(step(q, k, v, mask), want2) and step.captures == 2
    # a third key evicts the least recently used entry (max_graphs = 2) — and everything still answers correctly
    k3, v3 = k[:, :, :n2 - 64], v[:, :, :n2 - 64]
    m3 = mask[..., :n2 - 64] if mask is not None else None
    assert torch.equal(step(q, k3, v3, m3), ffpa_attn_func(q, k3, v3, attn_mask=m3, enable_gqa=sh["gqa"])) and len(step) == 2 and step.captures == 3
    assert torch.equal(step(q, k2, v2, m2), want3) and step.captures == 4  # (it was the evicted one)
    step.clear()
    assert len(step) == 0


def test_decode_step_on_a_side_stream_and_inside_a_callers_capture(hip):
  from ffpa_attn_amd import DecodeStep, ffpa_attn_func

  q, k, v = _rand((1, 32, 1, 512), seed=81), _rand((1, 32, 2048, 512), seed=82), _rand((1, 32, 2048, 512), seed=83)
  want = ffpa_attn_func(q, k, v)
  step = DecodeStep()
  assert torch.equal(step(q, k, v), want)
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):
    o_side = step(q, k, v)  # another stream is another key (two replays of one graph on two streams would share its scratch)
  torch.cuda.current_stream().wait_stream(side)
  assert step.captures == 2 and torch.equal(o_side, want)
  # a caller capturing a graph of its own gets the plain call captured into ITS graph (no nested capture)
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    o_g = step(q, k, v)
  g.replay()
  torch.cuda.synchronize()
  assert step.captures == 2 and torch.equal(o_g, want)
  with pytest.raises(ValueError, match="dropout"):
    DecodeStep(dropout_p=0.1)


def test_deterministic_mode_makes_a_slice_batch_invariant_to_the_bit(hip, monkeypatch):
  """FFPA_HIP_DETERMINISTIC=1 (FFPA_FLAG_DETERMINISTIC): the bits of a (batch, head) slice do not depend on how many slices share the launch — checked where
  the default plan does depend on it: an under-filled prefill launch (2 heads split the KV axis, 32 heads do not) and a decode step (the split count aims at
  a workgroup count)."""
  from ffpa_attn_amd import ffpa_attn_func

  monkeypatch.setenv("FFPA_HIP_DETERMINISTIC", "1")
  q, k, v = _rand((1, 32, 1024, 512), seed=91), _rand((1, 32, 8192, 512), seed=92), _rand((1, 32, 8192, 512), seed=93)
  plan_few, plan_all = {}, {}
  hip.forward(q[:, :2], k[:, :2], v[:, :2], None, False, 512 ** -0.5, plan_out=plan_few)
  hip.forward(q, k, v, None, False, 512 ** -0.5, plan_out=plan_all)
  assert plan_few["splits"] == plan_all["splits"] == 1 and plan_few["block_rows"] == plan_all["block_rows"] == 128
  o_all = ffpa_attn_func(q, k, v)
  o_few = ffpa_attn_func(q[:, :2].contiguous(), k[:, :2].contiguous(), v[:, :2].contiguous())
  assert torch.equal(o_few, o_all[:, :2])
  qd = _rand((8, 32, 1, 512), seed=94)
  kd, vd = _rand((8, 32, 4096, 512), seed=95), _rand((8, 32, 4096, 512), seed=96)
  pd1, pd8 = {}, {}
  hip.forward(qd[:1, :1], kd[:1, :1], vd[:1, :1], None, False, 512 ** -0.5, plan_out=pd1)
  hip.forward(qd, kd, vd, None, False, 512 ** -0.5, plan_out=pd8)
  assert pd1["splits"] == pd8["splits"] == 4, (pd1, pd8)  # 64 tiles of 64 keys, 16 per range
  o8 = ffpa_attn_func(qd, kd, vd)
  o1 = ffpa_attn_func(qd[:1, :1].contiguous(), kd[:1, :1].contiguous(), vd[:1, :1].contiguous())
  assert torch.equal(o1, o8[:1, :1])
  # without the pin the under-filled launch splits (equal to rounding only)
  monkeypatch.delenv("FFPA_HIP_DETERMINISTIC")
  plan_def = {}
  hip.forward(q[:, :2], k[:, :2], v[:, :2], None, False, 512 ** -0.5, plan_out=plan_def)
  assert plan_def["splits"] > 1


def test_decode_step_with_a_device_side_kv_length(hip):
  """The serving pattern `DecodeStep` is made for: a static-capacity KV cache whose valid length lives ON THE DEVICE — a boolean key mask [1, 1, 1, capacity] and its
  key ranges (`HIPBackend(kv_bounds=...)`: [first, end, free_lo, free_hi) per 32-row block, a device tensor) — both updated in place between replays: one captured graph
  serves every length, the split kernel's workgroups past `end` leave at once (HBM bytes follow the valid length, not the capacity), and the result equals the plain
  call on the same arguments bit for bit and the call on the sliced cache to rounding."""
  from ffpa_attn_amd import DecodeStep, HIPBackend, ffpa_attn_func

  cap, D = 8192, 512
  q, k, v = _rand((1, 32, 1, D), seed=101), _rand((1, 32, cap, D), seed=102), _rand((1, 32, cap, D), seed=103)
  mask = torch.zeros(1, 1, 1, cap, dtype=torch.bool, device=q.device)
  bounds = torch.zeros(1, 1, 1, 4, dtype=torch.int32, device=q.device)
  backend = HIPBackend(forward=True, kv_bounds=bounds)
  step = DecodeStep(forward_backend=backend)

  def set_len(n):
    mask.zero_()
    mask[..., :n] = True
    bounds.copy_(torch.tensor([0, n, 0, n], dtype=torch.int32, device=q.device).view(1, 1, 1, 4))

  for n in (1500, 1501, 4096, 8192, 700):
    set_len(n)
    got = step(q, k, v, mask)
    want_same_args = ffpa_attn_func(q, k, v, attn_mask=mask, forward_backend=backend)
    want_sliced = ffpa_attn_func(q, k[:, :, :n], v[:, :, :n])
    assert torch.equal(got, want_same_args), n
    assert (got.float() - want_sliced.float()).abs().max().item() <= 2e-3, n
  assert step.captures == 1 and len(step) == 1  # one graph served five lengths
  # and the bytes follow the length: a step at 700 valid keys is several times cheaper than one at the full capacity
  def timed(n):
    set_len(n)
    for _ in range(5):
      step(q, k, v, mask)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
      step(q, k, v, mask)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / 50
  t_short, t_full = timed(700), timed(8192)
  assert t_short < 0.6 * t_full, (t_short, t_full)


def test_short_sequence_opt_in_routes_to_the_kernel_with_sdpas_answers(hip, monkeypatch):
  """FFPA_HIP_ALLOW_SHORT_SEQ=1 (functional.py `_allows_short_seq`): calls the reference's length thresholds hand to SDPA — 8 <= Nq < 512, Nkv < 512 — reach the HIP
  kernel instead, with the answers the default route gives (SDPA here): chunked prefill rows against a long context, plain and under a boolean mask and GQA; a short
  causal self-attention (Nq == Nkv: both causal conventions agree); `is_causal` with Nq != Nkv stays with SDPA (top-left there, tail-aligned here)."""
  import time

  import torch.nn.functional as F

  from ffpa_attn_amd import ffpa_attn_func

  calls = []
  real = hip.forward

  def counting(*a, **k):
    calls.append(1)
    return real(*a, **k)

  monkeypatch.setattr(hip, "forward", counting)
  g = torch.Generator(device="cuda").manual_seed(3)
  mk = lambda *shape: torch.randn(shape, dtype=torch.bfloat16, device="cuda", generator=g)  # noqa: E731
  q, k, v = mk(2, 8, 128, 512), mk(2, 2, 4096, 512), mk(2, 2, 4096, 512)
  mask = torch.rand((1, 1, 128, 4096), device="cuda", generator=g) > 0.3
  qs, ks, vs = mk(2, 8, 256, 512), mk(2, 8, 256, 512), mk(2, 8, 256, 512)
  cases = [("chunk vs context, GQA", (q, k, v), dict(enable_gqa=True)), ("... under a boolean mask", (q, k, v), dict(enable_gqa=True, attn_mask=mask)),
           ("short causal self-attention", (qs, ks, vs), dict(is_causal=True)), ("short keys", (qs, ks[:, :, :100], vs[:, :, :100]), dict())]
  monkeypatch.delenv("FFPA_HIP_ALLOW_SHORT_SEQ", raising=False)
  default = [ffpa_attn_func(*t, **kw) for _, t, kw in cases]
  assert not calls  # the reference's decisions: SDPA
  monkeypatch.setenv("FFPA_HIP_ALLOW_SHORT_SEQ", "1")
  for (name, t, kw), ref in zip(cases, default):
    n = len(calls)
    out = ffpa_attn_func(*t, **kw)
    assert len(calls) == n + 1, f"{name}: the kernel did not run"
    assert torch.allclose(out.float(), ref.float(), atol=2e-2, rtol=2e-2), f"{name}: {(out.float() - ref.float()).abs().max().item():.3e}"
    assert (out.float() - ref.float()).abs().max().item() <= 1e-2, name
  n = len(calls)
  out = ffpa_attn_func(q, k.repeat_interleave(4, 1), v.repeat_interleave(4, 1), is_causal=True)  # Nq != Nkv under is_causal: SDPA's top-left mask, SDPA's call
  assert len(calls) == n
  assert torch.equal(out, F.scaled_dot_product_attention(q, k.repeat_interleave(4, 1), v.repeat_interleave(4, 1), is_causal=True))
  # what the switch is for (informative: printed with -s)
  def timed(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
      fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

  t_k = timed(lambda: ffpa_attn_func(q, k, v, enable_gqa=True))
  monkeypatch.delenv("FFPA_HIP_ALLOW_SHORT_SEQ")
  t_s = timed(lambda: ffpa_attn_func(q, k, v, enable_gqa=True))
  print(f"SHORTSEQ B2 Hq8/Hkv2 Nq128 Nkv4096 D512: kernel {t_k:.3f} ms, default route (SDPA) {t_s:.3f} ms ({t_s / t_k:.1f} x)")
