"""Pin the oracle to EXECUTED reference code: run the reference's own Triton forward kernel on the CPU and keep its outputs.

    python tests/golden/make_triton_golden.py          (authoring container only: needs /root/reference and triton)

The reference's large-head-dim arithmetic that can run without an NVIDIA GPU is its Triton statement of the algorithm
(src/ffpa_attn/triton/_ffpa_fwd.py: kernel `_ffpa_fwd_kernel_impl` :302-495, launcher `_ffpa_attn_forward_generic_impl`
:863-1041, entry `_ffpa_attn_forward_impl` :1337-1462).  Under ``TRITON_INTERPRET=1`` triton (3.6.0 here) executes the
kernel's Python body with numpy on CPU tensors.  Two host-side device queries stand in the way and are stubbed HERE, in
this generator only — they choose a launch configuration, not arithmetic:

  * ``_get_decode_num_splits`` reads ``torch.cuda.get_device_properties`` (:268-273)  -> 1 (the generic, unsplit kernel);
  * ``lookup_persistent_config`` reads ``torch.cuda.current_device`` (_persistent_autotune.py:534) -> None, i.e. the
    launcher's built-in default tile (BLOCK_M 128, BLOCK_N 64, head-dim blocks of 64: :985-992).

What was found (recorded in DESIGN.md §4): float16 runs and agrees with fp32 math to 5e-4; bfloat16 does NOT run
correctly under the interpreter (numpy has no bf16: the interpreter computes on the raw uint16 patterns, outputs ~1e8), so
the executed-reference pin is float16-only.  The kernel template, the oracle and the HIP kernel treat both 16-bit dtypes
with the same code, differing in the rounding of P and O alone.

Nothing of the reference travels: the fixture holds OUTPUTS only (O in fp16 bits, LSE fp32).  Inputs are re-created by
``triton_cases.triton_case_inputs`` from numpy's frozen legacy generator (``RandomState``), identically in the tests.
"""

from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from triton_cases import CASES, triton_case_inputs  # the input recipe shared with the tests (tests/golden/triton_cases.py)


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
  store, meta = {}, []
  for case in CASES:
    name, B, Hq, Hkv, Nq, Nkv, D, causal, bshape = case
    q, k, v, bias = (None if a is None else torch.from_numpy(a) for a in triton_case_inputs(case))
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
    print(f"{name}: max |O_triton - fp32 math| = {err:.2e}, max |LSE - ref| = {lerr:.2e}")
    assert err < 3e-3 and lerr < 1e-3, name
    store[f"{name}.o"] = o.contiguous().view(torch.int16).numpy().view(np.uint16).copy()
    store[f"{name}.lse"] = lse[..., :Nq].contiguous().numpy()
    meta.append({"name": name, "B": B, "Hq": Hq, "Hkv": Hkv, "Nq": Nq, "Nkv": Nkv, "D": D, "causal": causal, "bias_shape": bshape, "dtype": "fp16"})
  np.savez_compressed(os.path.join(HERE, "ref_triton_cases.npz"), **store)
  with open(os.path.join(HERE, "ref_triton_cases.json"), "w") as f:
    json.dump({"source": f"reference src/ffpa_attn/triton/_ffpa_fwd.py::_ffpa_attn_forward_impl under TRITON_INTERPRET=1, triton {triton.__version__}, "
                         f"torch {torch.__version__}; default tile BLOCK_M=128 BLOCK_N=64 head-dim blocks 64", "cases": meta}, f, indent=1)
  print("wrote ref_triton_cases.{npz,json}")


if __name__ == "__main__":
  main()
