"""Pin the oracle to EXECUTED reference code: run the reference's own Triton forward kernel on the CPU and keep its outputs.

    python tests/golden/make_triton_golden.py [--only NAME,NAME]     (authoring container only: needs /root/reference and triton;
                                                                      --only: run these cases and merge them into the existing fixture)
    python tests/golden/make_triton_golden.py --dropout              (the dropout cases only -> ref_triton_dropout.{npz,json})
    python tests/golden/make_triton_golden.py --decode               (the split-KV decode cases only -> ref_triton_decode.{npz,json})
    python tests/golden/make_triton_golden.py --api                  (the reference's argument handling + its kernel -> ref_triton_api.{npz,json})

The reference's large-head-dim arithmetic that can run without an NVIDIA GPU is its Triton statement of the algorithm
(src/ffpa_attn/triton/_ffpa_fwd.py: kernel `_ffpa_fwd_kernel_impl` :302-495, launcher `_ffpa_attn_forward_generic_impl`
:863-1041, entry `_ffpa_attn_forward_impl` :1337-1462).  Under ``TRITON_INTERPRET=1`` triton (3.6.0 here) executes the
kernel's Python body with numpy on CPU tensors.  Two host-side device queries stand in the way and are stubbed HERE, in
this generator only — they choose a launch configuration, not arithmetic:

  * ``_get_decode_num_splits`` reads ``torch.cuda.get_device_properties`` (:268-273)  -> 1 (the generic, unsplit kernel);
  * ``lookup_persistent_config`` reads ``torch.cuda.current_device`` (_persistent_autotune.py:534) -> None, i.e. the
    launcher's built-in default tile (BLOCK_M 128, BLOCK_N 64, head-dim blocks of 64: :985-992).

bfloat16: triton's interpreter keeps bf16 tensors as their uint16 storage bits (numpy has no bf16) and converts correctly only in
explicit casts; its ``create_dot`` multiplies the raw bit patterns as integers (outputs ~1e8) and its fp32 -> bf16 cast truncates
instead of rounding to nearest even.  Both are gaps of the INTERPRETER, not of the reference kernel, and are closed HERE, in this
generator only, by wrapping two interpreter functions (the reference source is imported unchanged):

  * ``InterpreterBuilder.create_dot``: bf16 operands are widened to fp32 first (exact) — what the matrix unit does;
  * ``_convert_float`` for fp32 -> bf16: round to nearest even (``tl.Tensor.to`` on hardware), bf16 -> fp32: bits << 16.

With that the reference's kernel body executes on bf16 inputs with hardware semantics: bf16 products accumulated in fp32, P and O
rounded to bf16 (RTNE).  The fp16 cases do not touch either wrapper.

Nothing of the reference travels: the fixture holds OUTPUTS only (O in fp16 / bf16 storage bits, LSE fp32).  Inputs are re-created by
``triton_cases.triton_case_inputs`` from numpy's frozen legacy generator (``RandomState``), identically in the tests.
"""

from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from triton_cases import API_CASES, CASES, DECODE_CASES, DROPOUT_CASES, api_case_inputs, bf16_bits_to_f32, f32_to_bf16_bits, triton_case_inputs  # the input recipe shared with the tests


def main():
  os.environ["TRITON_INTERPRET"] = "1"  # before triton is imported
  sys.path.insert(0, "/root/reference/src")
  import torch
  import triton

  try:
    from ffpa_attn.triton import _ffpa_fwd as ref
  except ImportError:
    sys.exit("the reference is only available in the authoring container (/root/reference)")
  ref._get_decode_num_splits = lambda *a, **k: 1     # device query -> the generic kernel
  ref.lookup_persistent_config = lambda req: None    # device query -> the launcher's default tile

  # ---- the interpreter's bf16 gaps (see the module docstring); nothing in the reference is touched
  import triton.language as tl
  import triton.runtime.interpreter as ti

  orig_convert, orig_dot = ti._convert_float, ti.InterpreterBuilder.create_dot

  def convert(inp, in_dt, out_dt, rounding_mode):
    if in_dt == tl.float32 and out_dt == tl.bfloat16:
      return f32_to_bf16_bits(np.ascontiguousarray(inp, dtype=np.float32))
    if in_dt == tl.bfloat16 and out_dt == tl.float32:
      return bf16_bits_to_f32(np.ascontiguousarray(inp).view(np.uint16)).view(np.uint32)
    return orig_convert(inp, in_dt, out_dt, rounding_mode)

  def dot(self, a, b, d, input_precision, max_num_imprecise_acc):
    if a.dtype.scalar == tl.bfloat16 or b.dtype.scalar == tl.bfloat16:
      af = bf16_bits_to_f32(a.data) if a.dtype.scalar == tl.bfloat16 else a.data.astype(np.float32)
      bf = bf16_bits_to_f32(b.data) if b.dtype.scalar == tl.bfloat16 else b.data.astype(np.float32)
      return ti.TensorHandle(np.matmul(af, bf, dtype=np.float32) + d.data, d.dtype.scalar)
    return orig_dot(self, a, b, d, input_precision, max_num_imprecise_acc)

  ti._convert_float = convert
  ti.InterpreterBuilder.create_dot = dot

  if "--dropout" in sys.argv:
    return make_dropout(ref, torch, triton)
  if "--decode" in sys.argv:
    return make_decode(ref, torch, triton)
  if "--api" in sys.argv:
    return make_api(ref, torch, triton)
  store, meta = {}, []
  only = None
  if "--only" in sys.argv:
    only = set(sys.argv[sys.argv.index("--only") + 1].split(","))
    old = np.load(os.path.join(HERE, "ref_triton_cases.npz"))
    store = {k_: old[k_] for k_ in old.files}
    meta = [m for m in json.load(open(os.path.join(HERE, "ref_triton_cases.json")))["cases"] if m["name"] not in only]
  for case in CASES:
    name, B, Hq, Hkv, Nq, Nkv, D, causal, bshape, dtype, spike = case
    if only is not None and name not in only:
      continue
    tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
    arrs = triton_case_inputs(case)
    q, k, v, bias = (None if a is None else (torch.from_numpy(a) if dtype == "fp16" else torch.from_numpy(a.view(np.int16)).view(tdt)) for a in arrs)
    o = torch.zeros_like(q)
    lse = torch.zeros(B, Hq, (Nq + 127) // 128 * 128, dtype=torch.float32)
    ref._ffpa_attn_forward_impl(q, k, v, o, lse, attn_bias=bias, causal=causal)
    # sanity: the interpreter really computed attention (fp32 math of the same inputs)
    g = Hq // Hkv
    s = (q.float() @ k.float().repeat_interleave(g, 1).transpose(-1, -2)) * D ** -0.5
    if bias is not None:
      s = s + bias.float()
    if causal:
      r, c = torch.arange(Nq)[:, None], torch.arange(Nkv)[None, :]
      s = s.masked_fill(c > r + (Nkv - Nq), float("-inf"))
    want = torch.softmax(s, -1) @ v.float().repeat_interleave(g, 1)
    err = (o.float() - want).abs().max().item()
    lerr = (lse[..., :Nq] - torch.logsumexp(s, -1)).abs().max().item()
    print(f"{name} [{dtype}]: max |O_triton - fp32 math| = {err:.2e}, max |LSE - ref| = {lerr:.2e}", flush=True)
    assert err < (3e-3 if dtype == "fp16" else 2e-2) and lerr < 1e-3, name
    store[f"{name}.o"] = o.contiguous().view(torch.int16).numpy().view(np.uint16).copy()
    store[f"{name}.lse"] = lse[..., :Nq].contiguous().numpy()
    meta.append({"name": name, "B": B, "Hq": Hq, "Hkv": Hkv, "Nq": Nq, "Nkv": Nkv, "D": D, "causal": causal, "bias_shape": bshape, "dtype": dtype,
                 "late_spikes": spike})
  order = {c[0]: i for i, c in enumerate(CASES)}
  meta.sort(key=lambda m: order[m["name"]])
  np.savez_compressed(os.path.join(HERE, "ref_triton_cases.npz"), **store)
  with open(os.path.join(HERE, "ref_triton_cases.json"), "w") as f:
    json.dump({"source": f"reference src/ffpa_attn/triton/_ffpa_fwd.py::_ffpa_attn_forward_impl under TRITON_INTERPRET=1, triton {triton.__version__}, "
                         f"torch {torch.__version__}; default tile BLOCK_M=128 BLOCK_N=64 head-dim blocks 64", "cases": meta}, f, indent=1)
  print("wrote ref_triton_cases.{npz,json}")


def make_dropout(ref, torch, triton):
  """The reference's Triton forward with dropout_p > 0 (its Philox mapping is the CUDA kernels': triton/_ffpa_fwd.py:80-123).  Sanity: the
  executed output must differ from plain attention (dropout really happened) and be an unbiased estimate of it (mean over elements)."""
  store, meta = {}, []
  for case in DROPOUT_CASES:
    name, B, Hq, Hkv, Nq, Nkv, D, causal, bshape, dtype, _, p, seed, offset = case
    tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
    arrs = triton_case_inputs(case)
    q, k, v, bias = (None if a is None else (torch.from_numpy(a) if dtype == "fp16" else torch.from_numpy(a.view(np.int16)).view(tdt)) for a in arrs)
    o = torch.zeros_like(q)
    lse = torch.zeros(B, Hq, (Nq + 127) // 128 * 128, dtype=torch.float32)
    ref._ffpa_attn_forward_impl(q, k, v, o, lse, attn_bias=bias, causal=causal, dropout_p=p, philox_seed=seed, philox_offset=offset)
    g = Hq // Hkv
    s = (q.float() @ k.float().repeat_interleave(g, 1).transpose(-1, -2)) * D ** -0.5
    if bias is not None:
      s = s + bias.float()
    if causal:
      r, c = torch.arange(Nq)[:, None], torch.arange(Nkv)[None, :]
      s = s.masked_fill(c > r + (Nkv - Nq), float("-inf"))
    plain = torch.softmax(s, -1) @ v.float().repeat_interleave(g, 1)
    diff = (o.float() - plain)
    lerr = (lse[..., :Nq] - torch.logsumexp(s, -1)).abs().max().item()
    print(f"{name} [{dtype}] p={p}: max |O_dropout - O_plain| = {diff.abs().max().item():.2e} (mean {diff.mean().item():+.1e}), max |LSE - ref| = {lerr:.2e}", flush=True)
    assert diff.abs().max().item() > 0.05 and abs(diff.mean().item()) < 0.02 and lerr < 1e-3, name  # LSE is the undropped one
    store[f"{name}.o"] = o.contiguous().view(torch.int16).numpy().view(np.uint16).copy()
    store[f"{name}.lse"] = lse[..., :Nq].contiguous().numpy()
    meta.append({"name": name, "B": B, "Hq": Hq, "Hkv": Hkv, "Nq": Nq, "Nkv": Nkv, "D": D, "causal": causal, "bias_shape": bshape, "dtype": dtype,
                 "dropout_p": p, "philox_seed": seed, "philox_offset": offset})
  np.savez_compressed(os.path.join(HERE, "ref_triton_dropout.npz"), **store)
  with open(os.path.join(HERE, "ref_triton_dropout.json"), "w") as f:
    json.dump({"source": f"reference src/ffpa_attn/triton/_ffpa_fwd.py::_ffpa_attn_forward_impl(dropout_p, philox_seed, philox_offset) under TRITON_INTERPRET=1, "
                         f"triton {triton.__version__}, torch {torch.__version__}", "cases": meta}, f, indent=1)
  print("wrote ref_triton_dropout.{npz,json}")


def make_decode(ref, torch, triton):
  """The reference's split-KV decode path (stage 1 per KV chunk + stage 2 LSE merge, triton/_ffpa_fwd.py:497-861) with the split count forced per case:
  `_get_decode_num_splits` reads the device (:268-273) and is replaced by the case's constant — a launch parameter, not arithmetic."""
  store, meta = {}, []
  for case in DECODE_CASES:
    name, B, Hq, Hkv, Nq, Nkv, D, causal, bshape, dtype, _, splits = case
    ref._get_decode_num_splits = lambda *a, _n=splits, **k: _n
    tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
    arrs = triton_case_inputs(case)
    q, k, v, bias = (None if a is None else (torch.from_numpy(a) if dtype == "fp16" else torch.from_numpy(a.view(np.int16)).view(tdt)) for a in arrs)
    o = torch.zeros_like(q)
    lse = torch.zeros(B, Hq, (Nq + 127) // 128 * 128, dtype=torch.float32)
    ref._ffpa_attn_forward_impl(q, k, v, o, lse, attn_bias=bias, causal=causal)
    g = Hq // Hkv
    s = (q.float() @ k.float().repeat_interleave(g, 1).transpose(-1, -2)) * D ** -0.5
    if bias is not None:
      s = s + bias.float()
    if causal:
      r, c = torch.arange(Nq)[:, None], torch.arange(Nkv)[None, :]
      s = s.masked_fill(c > r + (Nkv - Nq), float("-inf"))
    want = torch.softmax(s, -1) @ v.float().repeat_interleave(g, 1)
    err = (o.float() - want).abs().max().item()
    lerr = (lse[..., :Nq] - torch.logsumexp(s, -1)).abs().max().item()
    print(f"{name} [{dtype}] {splits} splits: max |O_triton - fp32 math| = {err:.2e}, max |LSE - ref| = {lerr:.2e}", flush=True)
    assert err < (3e-3 if dtype == "fp16" else 2e-2) and lerr < 1e-3, name
    store[f"{name}.o"] = o.contiguous().view(torch.int16).numpy().view(np.uint16).copy()
    store[f"{name}.lse"] = lse[..., :Nq].contiguous().numpy()
    meta.append({"name": name, "B": B, "Hq": Hq, "Hkv": Hkv, "Nq": Nq, "Nkv": Nkv, "D": D, "causal": causal, "bias_shape": bshape, "dtype": dtype,
                 "num_splits": splits})
  np.savez_compressed(os.path.join(HERE, "ref_triton_decode.npz"), **store)
  with open(os.path.join(HERE, "ref_triton_decode.json"), "w") as f:
    json.dump({"source": f"reference src/ffpa_attn/triton/_ffpa_fwd.py::_ffpa_attn_forward_impl -> _ffpa_attn_forward_decode_impl (split count forced) under "
                         f"TRITON_INTERPRET=1, triton {triton.__version__}, torch {torch.__version__}", "cases": meta}, f, indent=1)
  print("wrote ref_triton_decode.{npz,json}")


def make_api(ref, torch, triton):
  """The reference's own argument handling in front of its kernel: FFPAAttnMeta.from_kwargs().normalize(...) (functional.py:726-942, imported unchanged, CPU
  tensors) gives the validated inputs, the additive 4-D bias and the resolved scale; `_ffpa_attn_forward_impl` runs on exactly those."""
  from ffpa_attn.functional import FFPAAttnMeta

  store, meta_out = {}, []
  for case in API_CASES:
    name, B, Hq, Hkv, Nq, Nkv, D, causal, kind, dtype, scale, gqa = case
    tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
    qa, ka, va, mask_np = api_case_inputs(case)
    q, k, v = (torch.from_numpy(a) if dtype == "fp16" else torch.from_numpy(a.view(np.int16)).view(tdt) for a in (qa, ka, va))
    mask = None if mask_np is None else torch.from_numpy(mask_np)
    meta = FFPAAttnMeta.from_kwargs()
    meta, q2, k2, v2, bias = meta.normalize(q, k, v, mask, 0.0, causal, scale, gqa)
    assert not meta.fallback(q2, k2, mask, 0.0), name  # the shapes are ones the reference serves with its own kernels
    o = torch.zeros_like(q2)
    lse = torch.zeros(B, Hq, (Nq + 127) // 128 * 128, dtype=torch.float32)
    ref._ffpa_attn_forward_impl(q2, k2, v2, o, lse, attn_bias=bias, causal=causal, softmax_scale=meta.attn_meta.scale)
    g = Hq // Hkv
    s = (q.float() @ k.float().repeat_interleave(g, 1).transpose(-1, -2)) * meta.attn_meta.scale
    if mask is not None:
      m4 = mask if mask.dim() == 4 else (mask.view(1, 1, Nq, Nkv) if mask.dim() == 2 else mask.view(B, 1, Nq, Nkv))
      s = s.masked_fill(~m4, float("-inf")) if mask.dtype == torch.bool else s + m4
    if causal:
      r, c = torch.arange(Nq)[:, None], torch.arange(Nkv)[None, :]
      s = s.masked_fill(c > r + (Nkv - Nq), float("-inf"))
    want = torch.softmax(s, -1) @ v.float().repeat_interleave(g, 1)
    err = (o.float() - want).abs().max().item()
    lerr = (lse[..., :Nq] - torch.logsumexp(s, -1)).abs().max().item()
    print(f"{name} [{dtype}] scale {meta.attn_meta.scale:.5f}: max |O - fp32 math| = {err:.2e}, max |LSE - ref| = {lerr:.2e}", flush=True)
    assert err < (3e-3 if dtype == "fp16" else 2e-2) and lerr < 1e-3, name
    store[f"{name}.o"] = o.contiguous().view(torch.int16).numpy().view(np.uint16).copy()
    store[f"{name}.lse"] = lse[..., :Nq].contiguous().numpy()
    meta_out.append({"name": name, "B": B, "Hq": Hq, "Hkv": Hkv, "Nq": Nq, "Nkv": Nkv, "D": D, "is_causal": causal, "mask": kind, "dtype": dtype, "scale": scale,
                     "enable_gqa": gqa, "resolved_scale": meta.attn_meta.scale})
  np.savez_compressed(os.path.join(HERE, "ref_triton_api.npz"), **store)
  with open(os.path.join(HERE, "ref_triton_api.json"), "w") as f:
    json.dump({"source": f"reference FFPAAttnMeta.normalize (src/ffpa_attn/functional.py) -> triton/_ffpa_fwd.py::_ffpa_attn_forward_impl under TRITON_INTERPRET=1, "
                         f"triton {triton.__version__}, torch {torch.__version__}", "cases": meta_out}, f, indent=1)
  print("wrote ref_triton_api.{npz,json}")


if __name__ == "__main__":
  main()
