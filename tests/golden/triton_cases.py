"""Inputs of the executed-reference fixtures (tests/golden/ref_triton_cases.npz): the ONE recipe shared by the generator
(make_triton_golden.py, authoring container) and the tests (CPU oracle test, GPU kernel test).  The fixture holds outputs
only; q / k / v / bias are re-created here from numpy's frozen legacy generator."""

import numpy as np

# name, B, Hq, Hkv, Nq, Nkv, D, causal (tail-aligned, the reference's convention), bias shape or None, dtype, late spikes
CASES = [
  ("d320_tail", 1, 2, 2, 200, 333, 320, False, None, "fp16", False),
  ("d512_gqa_causal", 1, 2, 1, 192, 320, 512, True, None, "fp16", False),
  ("d1024_bias", 1, 1, 1, 130, 257, 1024, False, (1, 1, 130, 257), "fp16", False),
  ("d512_keybias_tail", 2, 2, 2, 72, 777, 512, False, (2, 1, 1, 777), "fp16", False),
  # a long row with late score spikes: keys 900 and 1400 are scaled up so that the running row max jumps by far more than the
  # native kernel's lazy-rescale threshold late in the KV walk (the reference's Triton statement rescales every step)
  ("d512_late_spike", 1, 1, 1, 128, 1536, 512, False, None, "fp16", True),
  # bfloat16 — the dtype of every BASELINE config
  ("bf16_d512_tail", 1, 2, 2, 640, 2049, 512, False, None, "bf16", False),
  ("bf16_d320_gqa_causal", 1, 4, 2, 192, 320, 320, True, None, "bf16", False),
  ("bf16_d1024_bias", 1, 1, 1, 130, 257, 1024, False, (1, 1, 130, 257), "bf16", False),
  ("bf16_d512_keybias_spike", 2, 2, 2, 72, 1100, 512, False, (2, 1, 1, 1100), "bf16", True),
  # the BASELINE key count (round 4): one head, 128 query rows against 8192 keys at D = 512 — the reference's own arithmetic at the headline's
  # row length (its suite goes to 8191 / 8192-key tails: tests/test_ffpa_fwd.py:1110-1143); non-causal and tail-aligned causal
  ("bf16_d512_n8192", 1, 1, 1, 128, 8192, 512, False, None, "bf16", False),
  ("bf16_d512_n8192_causal", 1, 1, 1, 128, 8192, 512, True, None, "bf16", False),
]

# Dropout (round 4): the reference's Triton forward carries the SAME Philox mapping as its CUDA kernels (triton/_ffpa_fwd.py:80-123: element offset =
# philox_offset + ((b Hq + h) Nq + r) Nkv + k, keep iff u > p, P scaled by 1 / (1 - p) after the row sum), so its executed outputs pin the oracle's —
# and through it the HIP kernel's — dropout stream, offsets and scaling.  Same tuple as CASES + (dropout_p, philox_seed, philox_offset); outputs in
# ref_triton_dropout.npz.  Offsets that are not multiples of 4 start inside a Philox quad; B = 2 / Hq = 2 exercise the (b, h) term.
DROPOUT_CASES = [
  ("drop_fp16_d320_gqa", 1, 2, 1, 70, 200, 320, False, None, "fp16", False, 0.3, 0x1234567890ABCDEF & (2 ** 62 - 1), 1003),
  ("drop_bf16_d512_causal", 1, 2, 2, 130, 257, 512, True, None, "bf16", False, 0.1, 42, 0),
  ("drop_bf16_d1024_keybias", 2, 1, 1, 40, 300, 1024, False, (2, 1, 1, 300), "bf16", False, 0.5, 2 ** 40 + 7, 4 * 12345 + 2),
]

# Split-KV decode (round 4): the reference's Triton stage-1 (per-chunk online softmax, partial O_c = acc / l_c, LSE_c) + stage-2 (LSE merge) kernels
# (triton/_ffpa_fwd.py:497-861), executed with the split count forced (its heuristic reads the device): the semantics of SURVEY.md appendix A.7.  Same tuple as
# CASES + (num_splits,); outputs in ref_triton_decode.npz.  Nq > 1 runs the tl.dot tiles (fp32 products: agreement at output rounding); Nq = 1 runs the
# reference's GEMV branch, which multiplies q and k in the 16-bit dtype before summing (:613: one rounding per product — a property of that Triton statement,
# not of the CUDA split_kv.cuh path, which converts to fp32 first), so that case agrees to the product-rounding noise only (fp16; the bound is in the test).
DECODE_CASES = [
  ("decode_fp16_d320_nq7_causal_s3", 1, 4, 4, 7, 1111, 320, True, None, "fp16", False, 3),
  ("decode_bf16_d512_nq4_gqa_s4", 2, 8, 2, 4, 1500, 512, False, None, "bf16", False, 4),
  ("decode_fp16_d512_nq1_gqa_s4", 2, 8, 2, 1, 1500, 512, False, None, "fp16", False, 4),
  ("decode_bf16_d1024_nq2_keybias_s5", 1, 2, 1, 2, 700, 1024, False, (1, 1, 1, 700), "bf16", False, 5),
]

# Through the reference's PUBLIC argument handling (round 4): FFPAAttnMeta.normalize (functional.py:726-942: validation, default scale, enable_gqa,
# bool mask -> 0 / -inf in q.dtype, 2-D / 3-D masks -> broadcasting 4-D views) feeding its Triton forward — what `ffpa_attn_func(q, k, v, attn_mask=...,
# scale=..., enable_gqa=...)` computes at head dims > 256 and sequence lengths >= 512 (shorter ones fall back to SDPA: functional.py:717-724).
# (name, B, Hq, Hkv, Nq, Nkv, D, is_causal, user-mask kind, dtype, scale or None, enable_gqa); outputs in ref_triton_api.npz.
API_CASES = [
  ("api_fp16_d320_bool2d_gqa", 1, 4, 2, 512, 640, 320, False, "bool2d", "fp16", None, True),
  ("api_bf16_d512_bool3d_scale", 2, 2, 2, 520, 777, 512, False, "bool3d", "bf16", 0.05, False),
  ("api_bf16_d512_f32_keys_heads", 1, 2, 2, 512, 512, 512, False, "f32_heads_keys", "bf16", None, False),
  ("api_bf16_d1024_causal_gqa", 1, 2, 1, 512, 600, 1024, True, None, "bf16", None, True),
]


def api_case_inputs(case):
  """q, k, v as triton_case_inputs gives them + the USER's attn_mask (numpy bool / float32, in the shape the user passes) or None."""
  name, B, Hq, Hkv, Nq, Nkv, D, causal, kind, dtype, scale, gqa = case
  q, k, v, _ = triton_case_inputs((name, B, Hq, Hkv, Nq, Nkv, D, causal, None, dtype, False))
  rs = np.random.RandomState((abs(hash_name(name)) + 17) % (2 ** 31))
  mask = None
  if kind == "bool2d":
    mask = rs.random_sample((Nq, Nkv)) < 0.7
    mask[:, 0] = True  # never a whole row hidden
  elif kind == "bool3d":
    mask = rs.random_sample((B, Nq, Nkv)) < 0.5
    mask[..., 3] = True
  elif kind == "f32_heads_keys":
    mask = (rs.standard_normal((1, Hq, 1, Nkv)) * 0.7).astype(np.float32)
  return q, k, v, mask


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
  """fp32 -> bfloat16 storage bits, round to nearest even (NaN kept quiet): what every bf16 store of the hardware does."""
  u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
  rounded = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
  nan = np.isnan(x)
  if nan.any():
    rounded = np.where(nan, np.uint16(0x7FC0), rounded)
  return rounded.reshape(np.shape(x))


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
  return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(np.shape(b))


def triton_case_inputs(case):
  """q, k, v (and additive bias) of a case as numpy arrays — float16 arrays for fp16 cases, uint16 bfloat16 STORAGE BITS for bf16
  cases.  The ONE recipe shared by the generator and the tests."""
  name, B, Hq, Hkv, Nq, Nkv, D, causal, bshape, dtype, spike = case[:11]
  rs = np.random.RandomState(abs(hash_name(name)) % (2 ** 31))
  q = rs.standard_normal((B, Hq, Nq, D)).astype(np.float32)
  k = rs.standard_normal((B, Hkv, Nkv, D)).astype(np.float32)
  v = rs.standard_normal((B, Hkv, Nkv, D)).astype(np.float32)
  if spike:
    k[:, :, 900 % Nkv] *= 6.0
    k[:, :, 1400 % Nkv] *= 9.0
  bias = None
  if bshape is not None:
    bias = (rs.standard_normal(bshape) * 0.5).astype(np.float32)
    hide = rs.random_sample(bshape) < 0.15  # some -inf entries, never a whole row
    hide[..., 0] = False
    bias = np.where(hide, np.float32(-np.inf), bias).astype(np.float32)
  if dtype == "fp16":
    cast = lambda a: None if a is None else a.astype(np.float16)  # noqa: E731
  else:
    cast = lambda a: None if a is None else f32_to_bf16_bits(a)  # noqa: E731
  return cast(q), cast(k), cast(v), cast(bias)


def to_f32(a, dtype):
  """The fp32 values of an input array of a case (float16 array or bf16 bits)."""
  return None if a is None else (a.astype(np.float32) if dtype == "fp16" else bf16_bits_to_f32(a))


def hash_name(name: str) -> int:
  h = 2166136261
  for ch in name.encode():
    h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
  return h
