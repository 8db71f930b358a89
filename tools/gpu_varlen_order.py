"""Packed batches: does the ORDER of the sequences in the batch matter?  (Inside an XCD the packed kernel walks the sequences in the caller's order, longest row
tiles first within a sequence: a long sequence late in the batch starts late.)  The bench batch in the caller's order, sorted longest-first, shortest-first, and
a few random permutations; same work, same kernel.  Developer tool."""
import itertools, os, sys, random
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ffpa_attn_amd import ffpa_attn_varlen_func


def timeit(fn, reps=40, warm=60):
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


def main():
  hq, hkv, d = 32, 8, 512
  base = [4096, 512, 2048, 1024, 3072, 256, 4864, 512]
  orders = {"caller": base, "longest first": sorted(base, reverse=True), "shortest first": sorted(base)}
  rnd = random.Random(0)
  for i in range(3):
    p = base[:]
    rnd.shuffle(p)
    orders[f"random {i}"] = p
  torch.manual_seed(0)
  t = sum(base)
  q = torch.randn(t, hq, d, dtype=torch.bfloat16, device="cuda")
  k = torch.randn(t, hkv, d, dtype=torch.bfloat16, device="cuda")
  v = torch.randn(t, hkv, d, dtype=torch.bfloat16, device="cuda")
  flops = 4 * hq * d * sum(n * (n + 1) // 2 for n in base)
  fns = {}
  for name, lens in orders.items():
    cu = torch.tensor([0, *np.cumsum(lens).tolist()], dtype=torch.int32, device="cuda")
    fns[name] = (lambda cu=cu, m=max(lens): ffpa_attn_varlen_func(q, k, v, cu, cu, m, m, causal=True, enable_gqa=True))
  res = {n: [] for n in fns}
  for _ in range(3):
    for n, f in fns.items():
      res[n].append(timeit(f))
  for n, ts in res.items():
    tm = sorted(ts)[1]
    print(f"VARLENORDER {n:15s} {orders[n]}: {tm * 1e3:8.1f} us {flops / tm / 1e9:7.1f} TF", flush=True)


if __name__ == "__main__":
  main()
