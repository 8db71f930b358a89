"""The launch plan of a random corpus of calls, asked of a library WITHOUT a GPU (`ffpa_attn_fwd_plan` / `_kernel` / `_workspace_bytes` price the MI355X figures
when no device is there) — to prove that a change of ffpa_capi.hip's plan code leaves every plan where it was:

    python tools/plan_corpus.py LIB [--n 6000] [--seed 0] > plans.txt      (one line per call; diff two libraries' outputs)

The corpus walks head dims, batch / head counts around the round and under-fill boundaries, short-query and prefill lengths, the causal flag with both
alignments, every bias kind and layout, dropout, forced split counts, the plan flags and faked CU counts (FFPA_HIP_FAKE_CUS = 128 / 256 / 304)."""
import argparse
import ctypes
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("lib")
  ap.add_argument("--n", type=int, default=6000)
  ap.add_argument("--seed", type=int, default=0)
  args = ap.parse_args()
  from ffpa_attn_amd import hip

  lib = hip.load_library(os.path.abspath(args.lib))
  rng = random.Random(args.seed)
  P = hip.FfpaFwdParams
  for i in range(args.n):
    cus = rng.choice(["", "", "128", "304"])
    if cus:
      os.environ["FFPA_HIP_FAKE_CUS"] = cus
    else:
      os.environ.pop("FFPA_HIP_FAKE_CUS", None)
    p = P()
    p.struct_size = ctypes.sizeof(P)
    p.abi_version = hip.ABI_VERSION
    p.q = p.k = p.v = p.o = 16
    d = rng.choice([64, 128, 192, 256, 320, 384, 448, 512, 576, 640, 768, 1024, 72, 328, 520])
    hkv = rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 16, 20, 32, 40])
    grp = rng.choice([1, 1, 1, 4, 8])
    b = rng.choice([1, 1, 1, 2, 3, 8])
    nq = rng.choice([1, 2, 4, 7, 16, 32, 33, 512, 1024, 2048, 4096, 8192, 8191, 640])
    nkv = rng.choice([512, 777, 1024, 2048, 4096, 8192, 16384, 32768, 140000])
    p.batch, p.heads_q, p.heads_kv, p.seqlen_q, p.seqlen_kv, p.head_dim = b, hkv * grp, hkv, nq, nkv, d
    for name, h, n in (("q_stride", hkv * grp, nq), ("k_stride", hkv, nkv), ("v_stride", hkv, nkv), ("o_stride", hkv * grp, nq)):
      getattr(p, name)[:] = [h * n * d, n * d, d]
    p.dtype = rng.choice([0, 1])
    p.causal = rng.choice([0, 0, 1])
    p.causal_offset = rng.choice([nkv - nq, 0]) if p.causal else 0
    p.softmax_scale = d ** -0.5
    p.rescale_threshold = -1.0
    p.dropout_p = rng.choice([0.0, 0.0, 0.0, 0.1])
    bias = rng.choice([None, None, "key", "dense", "bool", "boolkey", "f32", "unaligned"]) if not p.causal else None
    if bias:
      p.bias = 16 if bias != "unaligned" else 18
      p.bias_dtype = {"key": 2, "dense": 2, "bool": 4, "boolkey": 4, "f32": 3, "unaligned": 1}[bias]
      p.bias_stride[:] = [0, 0, 0 if bias in ("key", "boolkey") else nkv, 1]
      if bias in ("bool", "dense") and rng.random() < 0.5:
        p.kv_bounds = 16
    p.flags = rng.choice([0, 0, 0, 0, 0x1000, 0x2000, 0x40, 0x8])
    p.num_splits = rng.choice([0, 0, 0, 1, 2, 5]) if not (p.flags & 0x40) else rng.choice([2, 3, 7])
    if rng.random() < 0.85:
      p.workspace, p.workspace_bytes = 16, rng.choice([1 << 62, 1 << 62, 64 << 20])
    plan = (ctypes.c_int * 4)()
    rc = lib.ffpa_attn_fwd_plan(ctypes.byref(p), plan)
    name = ctypes.create_string_buffer(200)
    rk = lib.ffpa_attn_fwd_kernel(ctypes.byref(p), name, len(name)) if rc == 0 else -1
    ws = int(lib.ffpa_attn_fwd_workspace_bytes(ctypes.byref(p))) if rc == 0 else -1
    print(i, cus or "-", b, hkv * grp, hkv, nq, nkv, d, p.dtype, p.causal, p.causal_offset, bias, bool(p.kv_bounds), p.dropout_p > 0, hex(p.flags), p.num_splits, bool(p.workspace),
          "->", rc, list(plan), rk, name.value.decode(), ws)


if __name__ == "__main__":
  main()
