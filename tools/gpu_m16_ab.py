"""A/B of the unmasked D = 512 prefill kernel on the 16x16x32 MFMA shape vs the 32x32x16 build (developer tool):
results against each other and against fp32 math on small cases, then timing on the headline shapes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffpa_attn_amd import hip
from ffpa_attn_amd.flops import attention_fwd_flops


def ref(q, k, v, causal, off=None):
  g = q.size(1) // k.size(1)
  s = (q.float() @ k.float().repeat_interleave(g, 1).transpose(-1, -2)) * q.size(-1) ** -0.5
  if causal:
    nq, nk = q.size(2), k.size(2)
    off = nk - nq if off is None else off
    r, c = torch.arange(nq, device=q.device)[:, None], torch.arange(nk, device=q.device)[None, :]
    s = s.masked_fill(c > r + off, float("-inf"))
  return torch.softmax(s, -1) @ v.float().repeat_interleave(g, 1), torch.logsumexp(s, -1)


DIMS = [int(x) for x in os.environ.get("M16_DIMS", "512").split(",")]
bad = 0
for DD in DIMS:
 for dt in (torch.bfloat16, torch.float16):
  for (B, Hq, Hkv, Nq, Nkv, causal) in ((1, 2, 2, 128, 64, False), (1, 2, 1, 200, 333, False), (2, 4, 2, 384, 384, True), (1, 2, 2, 77, 1000, True), (1, 1, 1, 1024, 2048, False)):
    torch.manual_seed(Nq + Nkv)
    q = torch.randn(B, Hq, Nq, DD, dtype=dt, device="cuda")
    k = torch.randn(B, Hkv, Nkv, DD, dtype=dt, device="cuda")
    v = torch.randn(B, Hkv, Nkv, DD, dtype=dt, device="cuda")
    o1, l1 = hip.forward(q, k, v, None, causal, DD ** -0.5)
    o0, l0 = hip.forward(q, k, v, None, causal, DD ** -0.5, flags=hip.FLAG_NO_M16)
    ow, lw = ref(q, k, v, causal)
    e1, e0 = (o1.float() - ow).abs().max().item(), (o0.float() - ow).abs().max().item()
    el = (l1 - lw).abs().max().item()
    ok = e1 < 2 * max(e0, 2e-3) and el < 2e-3 and not torch.isnan(o1).any()
    bad += not ok
    print(f"M16 check D{DD} {str(dt)[6:]:>8} B{B} H{Hq}/{Hkv} Nq{Nq} Nkv{Nkv} causal{int(causal)}: |o16-ref| {e1:.2e}  |o32-ref| {e0:.2e}  |o16-o32| {(o1.float()-o0.float()).abs().max().item():.2e}  lse {el:.2e}  {'ok' if ok else 'BAD'}")
print("M16 correctness:", "PASS" if bad == 0 else f"FAIL ({bad})")
if bad == 0 or "--force" in sys.argv:
 for DD in DIMS:
  for (B, H, N, causal) in ((1, 32, 8192, False), (1, 32, 8192, True), (1, 32, 2048, False), (8, 32, 1024, False)):
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, DD, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    fl = attention_fwd_flops(B, H, N, N, DD) * (0.5 if causal else 1.0)
    res = {}
    for name, flags in (("m32", hip.FLAG_NO_M16), ("m16", 0), ("m32b", hip.FLAG_NO_M16), ("m16b", 0)):
      ts = []
      for rnd in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
          hip.forward(q, k, v, None, causal, DD ** -0.5, flags=flags, return_lse=False)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 5)
      res[name] = sorted(ts)[len(ts) // 2]
    print(f"M16 time D{DD} B{B} H{H} N{N} causal{int(causal)}: " + "  ".join(f"{n} {t:.4f} ms {fl / t / 1e9:.1f} TF" for n, t in res.items()))
