"""Generate the committed golden fixtures in tests/golden/ (run ONCE, in the authoring container).

    PYTHONPATH=/root/reference/src python tests/golden/make_golden.py

The reference package (xlite-dev/ffpa-attn, mounted read-only at /root/reference) is imported HERE
only: it never travels to the GPU box, in any form.  What it produces is data:

  dispatch_golden.json   FFPAAttnMeta.{from_kwargs,fallback,normalize} decisions of the reference on
                         meta tensors: per case either {"fallback": true}, {"raises": {type, match}}
                         or {"ffpa": {"scale": ..}} — pins this repo's host-side dispatch/validation.
  backend_golden.json    construction of the reference's Backend dataclasses with their full field lists (which kwargs construct, which assertion
                         trips, the fields afterwards) and where a config-2 call carrying the object is routed.
  flops_golden.json      reference attention_valid_pairs / attention_fwd_flops over a grid
                         (src/ffpa_attn/cli/_flops.py:15-53).
  cfg1_cpu.npz           BASELINE config 1 (B1 H4 N1024 D64 bf16, seed 0): q/k/v bits and the output of
                         the reference's ffpa_attn_func on CPU (it takes the SDPA fallback,
                         ffpa_attn_interface.py:165-176).
  small_cases.npz        small seeded cases (GQA, cross, causal offsets, masks, tails, D up to 1024):
                         inputs, PyTorch CPU SDPA output in the storage dtype (the reference's own test
                         oracle, tests/test_ffpa_fwd.py:48-51) and LSE from float64 math.
"""

from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def bits(t: torch.Tensor) -> np.ndarray:
  return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


# ------------------------------------------------------------------------------------ dispatch table
DISPATCH_CASES = []


def case(name, q, k, v=None, mask=None, dtype="bf16", **kw):
  DISPATCH_CASES.append({"name": name, "q": list(q), "k": list(k), "v": list(v or k), "mask": mask, "dtype": dtype, "kwargs": kw})


# BASELINE configs
case("cfg1", (1, 4, 1024, 64), (1, 4, 1024, 64))
case("cfg2", (1, 32, 8192, 512), (1, 32, 8192, 512))
case("cfg3", (1, 32, 8192, 1024), (1, 32, 8192, 1024))
case("cfg4_mask", (2, 32, 8192, 320), (2, 8, 2048, 320), mask={"shape": [8192, 2048], "dtype": "bool"}, enable_gqa=True)
case("cfg4_is_causal", (2, 32, 8192, 320), (2, 8, 2048, 320), enable_gqa=True, is_causal=True)
case("cfg5", (8, 32, 8192, 512), (8, 32, 8192, 512))
# head-dim boundaries
for d in (32, 64, 128, 256, 264, 320, 512, 640, 1000, 1024, 1032, 2048):
  case(f"d{d}", (1, 8, 1024, d), (1, 8, 1024, d))
# sequence-length boundaries (decode, short, tails)
for nq in (1, 7, 8, 15, 511, 512, 513, 5000):
  case(f"nq{nq}", (1, 8, nq, 512), (1, 8, 4096, 512))
for nkv in (1, 128, 511, 512, 8191):
  case(f"nkv{nkv}", (1, 8, 1024, 512), (1, 8, nkv, 512))
# error contracts (tests/test_ffpa_fwd.py:162-177,1146-1152,1199-1215,1366-1382)
case("fp32", (1, 8, 1024, 512), (1, 8, 1024, 512), dtype="fp32")
case("kv_seqlen_mismatch", (1, 8, 1024, 512), (1, 8, 1024, 512), v=(1, 8, 1023, 512))
case("heads_not_divisible", (1, 6, 1024, 512), (1, 4, 1024, 512), enable_gqa=True)
case("gqa_without_optin", (1, 8, 1024, 512), (1, 2, 1024, 512))
case("gqa_optin", (1, 8, 1024, 512), (1, 2, 1024, 512), enable_gqa=True)
case("causal_nkv_lt_nq", (1, 8, 2048, 512), (1, 8, 1024, 512), is_causal=True)
case("causal_ok", (1, 8, 1024, 512), (1, 8, 2048, 512), is_causal=True)
case("mask_and_causal", (1, 8, 1024, 512), (1, 8, 1024, 512), mask={"shape": [1024, 1024], "dtype": "bool"}, is_causal=True)
case("mask_bad_key_dim", (1, 8, 1024, 512), (1, 8, 1024, 512), mask={"shape": [1024, 1000], "dtype": "bool"})
case("mask_fp16_on_bf16", (1, 8, 1024, 512), (1, 8, 1024, 512), mask={"shape": [1, 1, 1, 1024], "dtype": "fp16"})
case("mask_fp32", (1, 8, 1024, 512), (1, 8, 1024, 512), mask={"shape": [1, 1, 1, 1024], "dtype": "fp32"})
case("mask_5d", (1, 8, 1024, 512), (1, 8, 1024, 512), mask={"shape": [1, 1, 1, 1, 1024], "dtype": "bool"})
case("dropout_1", (1, 8, 1024, 512), (1, 8, 1024, 512), dropout_p=1.0)
case("dropout_neg", (1, 8, 1024, 512), (1, 8, 1024, 512), dropout_p=-0.1)
case("scale_given", (1, 8, 1024, 512), (1, 8, 1024, 512), scale=0.125)
case("unknown_kwarg", (1, 8, 1024, 512), (1, 8, 1024, 512), foo=1)
case("bad_backend_str", (1, 8, 1024, 512), (1, 8, 1024, 512), backend="nope")
case("bad_backend_type", (1, 8, 1024, 512), (1, 8, 1024, 512), forward_backend=3)
case("sdpa_backend", (1, 8, 1024, 512), (1, 8, 1024, 512), backend="sdpa")
case("fwd_sdpa_backend", (1, 8, 1024, 512), (1, 8, 1024, 512), forward_backend="sdpa")
# backend-name contracts: the reference's CUDABackend is forward-only — CUDABackend() / backend="cuda" / backward_backend="cuda" trip its
# assertion (functional.py:266-268 with Backend.__post_init__ :189-196); forward_backend="cuda" is the supported spelling (:501-502)
case("backend_cuda_str", (1, 8, 1024, 512), (1, 8, 1024, 512), backend="cuda")
case("fwd_backend_cuda_str", (1, 8, 1024, 512), (1, 8, 1024, 512), forward_backend="cuda")
case("bwd_backend_cuda_str", (1, 8, 1024, 512), (1, 8, 1024, 512), backward_backend="cuda")
case("fwd_cuda_bwd_sdpa", (1, 8, 1024, 512), (1, 8, 1024, 512), forward_backend="cuda", backward_backend="sdpa")
case("backend_triton_str", (1, 8, 1024, 512), (1, 8, 1024, 512), backend="triton")
case("fwd_backend_triton_str", (1, 8, 1024, 512), (1, 8, 1024, 512), forward_backend="triton")
case("bwd_backend_sdpa_str", (1, 8, 1024, 512), (1, 8, 1024, 512), backward_backend="sdpa")
case("batch_mismatch", (2, 8, 1024, 512), (1, 8, 1024, 512))
case("headdim_mismatch", (1, 8, 1024, 512), (1, 8, 1024, 512), v=(1, 8, 1024, 256))

_DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32, "bool": torch.bool}


def run_dispatch():
  from ffpa_attn.functional import FFPAAttnMeta

  out = []
  for c in DISPATCH_CASES:
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
    rec = dict(c)
    try:
      meta = FFPAAttnMeta.from_kwargs(**kw)
      if meta.fallback(q, k, mask, dropout_p):
        rec["expect"] = {"fallback": True}
      else:
        meta, *_ = meta.normalize(q, k, v, mask, dropout_p, is_causal, scale, enable_gqa)
        rec["expect"] = {"ffpa": {"scale": meta.attn_meta.scale}}
    except Exception as e:  # noqa: BLE001 - recording the contract
      rec["expect"] = {"raises": {"type": type(e).__name__, "message": str(e)}}
    out.append(rec)
  return out


# ------------------------------------------------------------------------------------ backend objects
# Construction of the reference's Backend dataclasses (functional.py:176-470) with its full field lists: which kwargs construct, which trip
# which assertion, what the fields hold afterwards — and where a call carrying the object as forward_backend is routed (config-2 shapes on
# meta tensors: fused kernel or SDPA fallback).
BACKEND_CASES = [
  ("TritonBackend", {}), ("TritonBackend", {"forward": True}), ("TritonBackend", {"backward": True}),
  ("TritonBackend", {"autotune_mode": "max", "enable_tma": True}), ("TritonBackend", {"autotune": True, "enable_ws": True, "forward": True}),
  ("TritonBackend", {"autotune_mode": "slow"}), ("TritonBackend", {"persist_dkdv": True}), ("TritonBackend", {"persist_dkdv": True, "enable_tma": True}),
  ("TritonBackend", {"forward": True, "persist_dkdv": True, "enable_tma": True}), ("TritonBackend", {"forward": True, "split_launch": True}),
  ("TritonBackend", {"backward": True, "split_launch": True, "preprocess_d_chunk": True}), ("TritonBackend", {"backward": True, "grad_kv_storage_dtype": "fp32"}),
  ("TritonBackend", {"grad_q_storage_dtype": "fp16"}), ("TritonBackend", {"grad_q_storage_dtype": "bf16"}), ("TritonBackend", {"forward": True, "grad_kv_storage_dtype": "fp16"}),
  ("TritonBackend", {"foo": 1}),
  ("CUDABackend", {}), ("CUDABackend", {"forward": True}), ("CUDABackend", {"backward": False}), ("CUDABackend", {"forward": True, "fp8_smooth_k": False}),
  ("CUDABackend", {"forward": True, "enable_fp8": True}), ("CUDABackend", {"forward": True, "enable_fp4": True, "fp4_hybrid": False}),
  ("CUDABackend", {"forward": True, "enable_fp8": True, "enable_fp4": True}), ("CUDABackend", {"forward": True, "fp8_q_quant_method": "per_channel"}),
  ("CUDABackend", {"forward": True, "fp8_k_quant_method": "per_thread"}), ("CUDABackend", {"forward": True, "fp8_v_quant_method": "per_thread"}),
  ("CUDABackend", {"forward": True, "fp8_v_quant_method": "per_channel", "fp8_smooth_v": True}), ("CUDABackend", {"forward": True, "fp8_smooth_v": True}),
  ("CUDABackend", {"forward": True, "fp8_pv_acc_type": "bf16"}), ("CUDABackend", {"forward": True, "fp8_pv_acc_type": "f16", "fp8_qk_mm_type": "int8"}),
  ("CUDABackend", {"forward": True, "fp8_qk_mm_type": "fp4"}), ("CUDABackend", {"forward": True, "acc": "f64"}), ("CUDABackend", {"forward": True, "acc": "f16"}),
  ("CUDABackend", {"forward": True, "enable_tma": True}), ("CUDABackend", {"forward": True, "enable_cute": True, "stages": 2}),
  ("CUDABackend", {"forward": True, "enable_tma": True, "enable_cute": True, "enable_ws": True}),
  ("CUDABackend", {"forward": True, "fp8_hybrid": True, "fp8_hybrid_n_early": 128, "fp4_hybrid_n_early": 64}), ("CUDABackend", {"forward": True, "bar": 2}),
  ("CuTeDSLBackend", {}), ("CuTeDSLBackend", {"grad_kv_storage_dtype": "fp16"}), ("CuTeDSLBackend", {"forward": True, "grad_kv_storage_dtype": "fp16"}),
  ("CuTeDSLBackend", {"grad_kv_storage_dtype": "int8"}),
  ("SDPABackend", {}), ("SDPABackend", {"backward": True, "high_precision_grad": True}), ("SDPABackend", {"forward": True}),
]
# fields whose value depends on the reference's build / device (the CUDA backend resolves its pipeline depth from the GPU generation): not compared
BACKEND_DEVICE_FIELDS = ("stages",)


def _plain(v):
  return str(v) if isinstance(v, torch.dtype) else v


def run_backends():
  import dataclasses

  import ffpa_attn
  from ffpa_attn.functional import FFPAAttnMeta

  out = []
  for cls_name, kw in BACKEND_CASES:
    rec = {"cls": cls_name, "kwargs": kw}
    try:
      obj = getattr(ffpa_attn, cls_name)(**kw)
      rec["expect"] = {"fields": {f.name: _plain(getattr(obj, f.name)) for f in dataclasses.fields(obj) if f.name not in BACKEND_DEVICE_FIELDS}}
    except Exception as e:  # noqa: BLE001 - recording the contract
      rec["expect"] = {"raises": {"type": type(e).__name__, "message": str(e)}}
      out.append(rec)
      continue
    # where does a config-2 call carrying this object go?
    q = torch.empty((1, 32, 8192, 512), dtype=torch.bfloat16, device="meta")
    try:
      meta = FFPAAttnMeta.from_kwargs(**({"forward_backend": obj} if obj.forward else {"backward_backend": obj}))
      rec["route"] = {"fallback": bool(meta.fallback(q, q, None, 0.0))}
      if not rec["route"]["fallback"]:
        meta, *_ = meta.normalize(q, q, q, None, 0.0, True, None, False)
        rec["route"]["scale"] = meta.attn_meta.scale
        rec["route"]["fields_after"] = {f.name: _plain(getattr(meta.forward_meta, f.name)) for f in dataclasses.fields(meta.forward_meta)
                                        if f.name in ("is_causal", "fp8_hybrid", "fp4_hybrid")}
    except Exception as e:  # noqa: BLE001
      rec["route"] = {"raises": {"type": type(e).__name__, "message": str(e)}}
    out.append(rec)
  return out


# ------------------------------------------------------------------------------------ flops table
def run_flops():
  from ffpa_attn.cli._flops import attention_fwd_flops, attention_valid_pairs

  rows = []
  for (b, h, nq, nkv, d) in [(1, 32, 8192, 8192, 512), (1, 32, 8192, 8192, 1024), (2, 32, 8192, 2048, 320),
                             (8, 32, 8192, 8192, 512), (1, 4, 1024, 1024, 64), (1, 8, 1024, 8192, 512),
                             (1, 8, 1, 4096, 512), (1, 8, 129, 5000, 640), (2, 3, 1000, 1000, 320)]:
    for causal in (False, True):
      if causal and nkv < nq:
        continue
      rows.append({
        "B": b, "H": h, "Nq": nq, "Nkv": nkv, "D": d, "causal": causal,
        "pairs": int(attention_valid_pairs(nq, nkv, causal)),
        "flops": int(attention_fwd_flops(b, h, nq, nkv, d, causal)),
      })
  return rows


# ------------------------------------------------------------------------------------ tensors
def run_cfg1():
  from ffpa_attn import ffpa_attn_func

  torch.manual_seed(0)
  q = torch.randn(1, 4, 1024, 64, dtype=torch.bfloat16)
  k = torch.randn(1, 4, 1024, 64, dtype=torch.bfloat16)
  v = torch.randn(1, 4, 1024, 64, dtype=torch.bfloat16)
  o = ffpa_attn_func(q, k, v)
  o_sdpa = torch.nn.functional.scaled_dot_product_attention(q, k, v)
  assert torch.equal(o, o_sdpa), "reference config-1 path is expected to be the SDPA fallback"
  np.savez(os.path.join(HERE, "cfg1_cpu.npz"), q=bits(q), k=bits(k), v=bits(v), o_reference=bits(o))


SMALL = [
  # name, B, Hq, Hkv, Nq, Nkv, D, dtype, causal(tail-aligned), mask kind
  ("mha_d64", 1, 2, 2, 96, 160, 64, "bf16", False, None),
  ("gqa_cross_d320", 1, 4, 2, 70, 131, 320, "bf16", False, None),
  ("self_d512", 1, 1, 1, 130, 130, 512, "bf16", False, None),
  ("causal_self_d512", 1, 1, 1, 129, 129, 512, "bf16", True, None),
  ("causal_cross_tail_d320", 1, 2, 1, 65, 129, 320, "bf16", True, None),
  ("mqa_d1024", 1, 2, 1, 33, 97, 1024, "bf16", False, None),
  ("causal_d640", 1, 1, 1, 66, 144, 640, "bf16", True, None),
  ("boolmask_rowbcast_d512", 2, 1, 1, 40, 131, 512, "bf16", False, "bool_key"),
  ("addmask_full_d320", 1, 2, 2, 64, 100, 320, "bf16", False, "add_full"),
  ("addmask_f32_headbcast_d512", 2, 2, 1, 33, 97, 512, "bf16", False, "add_f32"),
  ("fp16_self_d512", 1, 1, 1, 96, 96, 512, "fp16", False, None),
  ("fp16_causal_d320", 1, 2, 2, 64, 100, 320, "fp16", True, None),
  ("d576_split", 1, 1, 1, 70, 97, 576, "bf16", False, None),
  ("topleft_causal_d320", 1, 2, 1, 130, 70, 320, "bf16", "topleft", None),
]


def run_small():
  store = {}
  meta = []
  for i, (name, B, Hq, Hkv, Nq, Nkv, D, dt, causal, mk) in enumerate(SMALL):
    torch.manual_seed(100 + i)
    tdt = _DT[dt]
    q = torch.randn(B, Hq, Nq, D, dtype=tdt)
    k = torch.randn(B, Hkv, Nkv, D, dtype=tdt)
    v = torch.randn(B, Hkv, Nkv, D, dtype=tdt)
    mask = None
    if mk == "bool_key":
      mask = torch.rand(B, 1, 1, Nkv) > 0.3
      mask[..., 0] = True
    elif mk == "add_full":
      mask = (torch.randn(1, Hq, Nq, Nkv) * 0.5).to(tdt)
    elif mk == "add_f32":
      mask = torch.randn(B, 1, Nq, Nkv, dtype=torch.float32) * 0.5
    sdpa_mask = mask
    causal_offset = None
    if causal == "topleft":
      causal_offset = 0
      sdpa_mask = torch.ones(Nq, Nkv, dtype=torch.bool).tril()
    elif causal:
      causal_offset = Nkv - Nq
      rows = torch.arange(Nq)[:, None]
      cols = torch.arange(Nkv)[None, :]
      sdpa_mask = cols <= rows + (Nkv - Nq)  # explicit tail-aligned mask, as tests/test_ffpa_fwd.py:1268-1304
    o = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=sdpa_mask, enable_gqa=(Hq != Hkv))
    # fp64 math for O and LSE
    g = Hq // Hkv
    qd, kd, vd = q.double(), k.double().repeat_interleave(g, 1), v.double().repeat_interleave(g, 1)
    s = qd @ kd.transpose(-1, -2) / (D ** 0.5)
    if sdpa_mask is not None:
      s = s + (torch.zeros_like(s).masked_fill(~sdpa_mask, float("-inf")) if sdpa_mask.dtype == torch.bool else sdpa_mask.double())
    lse = torch.logsumexp(s, dim=-1)
    store[f"{name}.q"], store[f"{name}.k"], store[f"{name}.v"] = bits(q), bits(k), bits(v)
    store[f"{name}.o_sdpa"] = bits(o)
    store[f"{name}.lse_f64"] = lse.numpy().astype(np.float32)
    if mask is not None:
      if mask.dtype == torch.bool:
        store[f"{name}.mask"] = mask.numpy()
      elif mask.dtype == torch.float32:
        store[f"{name}.mask"] = mask.numpy()
      else:
        store[f"{name}.mask_bits"] = bits(mask)
    meta.append({"name": name, "B": B, "Hq": Hq, "Hkv": Hkv, "Nq": Nq, "Nkv": Nkv, "D": D, "dtype": dt,
                 "causal": bool(causal), "causal_offset": causal_offset, "mask": mk})
  np.savez_compressed(os.path.join(HERE, "small_cases.npz"), **store)
  with open(os.path.join(HERE, "small_cases.json"), "w") as f:
    json.dump(meta, f, indent=1)


def main():
  try:
    import ffpa_attn  # noqa: F401
  except ImportError:
    sys.exit("run with PYTHONPATH=/root/reference/src (the reference is only available in the authoring container)")
  torch.set_num_threads(8)
  if "--backends-only" not in sys.argv:
    with open(os.path.join(HERE, "dispatch_golden.json"), "w") as f:
      json.dump(run_dispatch(), f, indent=1)
  with open(os.path.join(HERE, "flops_golden.json"), "w") as f:
    json.dump(run_flops(), f, indent=1)
  with open(os.path.join(HERE, "backend_golden.json"), "w") as f:
    json.dump(run_backends(), f, indent=1)
  if "--backends-only" in sys.argv:
    return
  run_cfg1()
  run_small()
  print("golden fixtures written to", HERE)


if __name__ == "__main__":
  main()
