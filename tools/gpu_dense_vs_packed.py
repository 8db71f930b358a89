"""Dense causal / GQA launches: the (batch, head, row tile) workgroup order against the head-chunk order (ffpa_capi.hip::pick_dense_head_chunk) — the same
call with and without FFPA_FLAG_NO_HEAD_CHUNKS, interleaved in one run — and, where the rule does not take the chunk order (MHA, non-causal), the same problem
through the packed-sequence call as one sequence per batch element (its kernel's order with chunks of one head).  Developer tool: profiles/r06_head_chunks.txt."""
import sys, os, math, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ffpa_attn_amd import ffpa_attn_varlen_func, hip


def timeit(fn, reps=30, warm=40):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps


torch.manual_seed(0)
SHAPES = ((1, 32, 8, 8192, 512, True), (1, 32, 32, 8192, 512, True), (2, 32, 8, 4096, 512, True), (1, 32, 8, 8192, 512, False), (1, 32, 8, 8192, 1024, True),
          (1, 32, 8, 8192, 320, True), (4, 32, 8, 2048, 512, True), (1, 64, 8, 8192, 512, True), (1, 32, 4, 8192, 512, True), (8, 32, 8, 1024, 512, True), (1, 32, 8, 16384, 512, True))
for (B, Hq, Hkv, N, D, causal) in SHAPES:
  q = torch.randn(B, N, Hq, D, dtype=torch.bfloat16, device="cuda").transpose(1, 2)
  k = torch.randn(B, N, Hkv, D, dtype=torch.bfloat16, device="cuda").transpose(1, 2)
  v = torch.randn(B, N, Hkv, D, dtype=torch.bfloat16, device="cuda").transpose(1, 2)
  plan = {}
  hip.forward(q, k, v, None, causal, D ** -0.5, plan_out=plan, return_lse=False)
  ruled = "head chunks" in plan["kernel"]
  base = lambda: hip.forward(q, k, v, None, causal, D ** -0.5, flags=hip.FLAG_NO_HEAD_CHUNKS, return_lse=False)[0]
  if ruled:
    other, what = (lambda: hip.forward(q, k, v, None, causal, D ** -0.5, return_lse=False)[0]), plan["kernel"].split("(")[1].rstrip(")")
  else:
    cu = torch.arange(0, (B + 1) * N, N, dtype=torch.int32, device="cuda")
    qp, kp, vp = (t.transpose(1, 2).reshape(B * N, t.size(1), D) for t in (q, k, v))
    other, what = (lambda: ffpa_attn_varlen_func(qp, kp, vp, cu, cu, N, N, causal=causal, enable_gqa=True).view(B, N, Hq, D).transpose(1, 2)), "not taken by the rule; the packed call"
  pairs = N * (N + 1) // 2 if causal else N * N
  fl = 4 * B * Hq * D * pairs
  res = [(timeit(base), timeit(other)) for _ in range(3)]
  tb, to = sorted(r[0] for r in res)[1], sorted(r[1] for r in res)[1]
  diff = (base().float() - other().float()).abs().max().item()
  print(f"HEADCHUNKS B{B} Hq{Hq}/Hkv{Hkv} N{N} D{D} causal={causal}: (batch, head, tile) order {tb * 1e3:8.1f} us {fl / tb / 1e9:7.1f} TF | {what}: {to * 1e3:8.1f} us {fl / to / 1e9:7.1f} TF ({tb / to:.3f} x) | maxdiff {diff:.1e}", flush=True)
