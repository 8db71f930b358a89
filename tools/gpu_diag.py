"""First-contact diagnostics for the HIP kernel on a real MI355X (developer tool, not a test).

Every case runs in its own subprocess with a timeout, so a fault or hang in one configuration does
not hide the others.  For each (D, shape, causal) it reports max |O - fp32 math| for the fast path
(LDS-DMA + ds_read_b64_tr_b16) and for the register-staged twin (FFPA_FLAG_DEBUG_SAFE_PATH), and
whether the two agree bit-for-bit — which isolates data-path problems from arithmetic ones.
"""

import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASE_SRC = r"""
import sys, json, torch
sys.path.insert(0, %(root)r)
from ffpa_attn_amd import hip
B,Hq,Hkv,Nq,Nkv,D,causal,dtype = %(case)r
dt = torch.bfloat16 if dtype == "bf16" else torch.float16
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(B,Hq,Nq,D,dtype=dt,device="cuda",generator=g)
k = torch.randn(B,Hkv,Nkv,D,dtype=dt,device="cuda",generator=g)
v = torch.randn(B,Hkv,Nkv,D,dtype=dt,device="cuda",generator=g)
scale = D ** -0.5
grp = Hq // Hkv
s = (q.float() @ k.float().repeat_interleave(grp,1).transpose(-1,-2)) * scale
if causal:
    r = torch.arange(Nq,device="cuda")[:,None]; c = torch.arange(Nkv,device="cuda")[None,:]
    s = s.masked_fill(c > r + (Nkv-Nq), float("-inf"))
ref = torch.softmax(s,-1) @ v.float().repeat_interleave(grp,1)
lse_ref = torch.logsumexp(s,-1)
res = {}
for name, flags in (("fast",0),("safe",hip.FLAG_DEBUG_SAFE_PATH)):
    try:
        o, lse = hip.forward(q,k,v,None,causal,scale,flags=flags)
        torch.cuda.synchronize()
        res[name] = {"max_err": (o.float()-ref).abs().max().item(), "lse_err": (lse-lse_ref).abs().max().item(),
                     "nan": int(torch.isnan(o).sum().item())}
        res[name+"_o"] = o
    except Exception as e:
        res[name] = {"error": str(e)[:300]}
if "fast_o" in res and "safe_o" in res:
    res["bit_equal"] = bool(torch.equal(res["fast_o"], res["safe_o"]))
    if not res["bit_equal"]:
        d = (res["fast_o"].float()-res["safe_o"].float()).abs()
        res["fast_vs_safe_max"] = d.max().item()
        bad = (d > 0).nonzero()
        res["first_bad"] = bad[:4].tolist(); res["n_bad"] = int(bad.shape[0])
res.pop("fast_o",None); res.pop("safe_o",None)
print("RESULT " + json.dumps(res))
"""

CASES = [
  # B, Hq, Hkv, Nq, Nkv, D, causal, dtype
  (1, 1, 1, 128, 64, 64, False, "bf16"),
  (1, 1, 1, 128, 128, 128, False, "bf16"),
  (1, 2, 1, 200, 333, 128, True, "bf16"),
  (1, 2, 2, 128, 256, 320, False, "bf16"),
  (1, 2, 2, 128, 256, 512, False, "bf16"),
  (2, 4, 2, 300, 700, 512, True, "bf16"),
  (1, 2, 2, 64, 128, 640, False, "bf16"),
  (1, 2, 2, 128, 256, 1024, False, "bf16"),
  (1, 2, 1, 190, 515, 1024, True, "bf16"),
  (1, 2, 2, 128, 256, 512, False, "fp16"),
]

PERF_SRC = r"""
import sys, json, time, torch
sys.path.insert(0, %(root)r)
from ffpa_attn_amd import hip
B,H,N,D = %(shape)r
torch.manual_seed(0)
q = torch.randn(B,H,N,D,dtype=torch.bfloat16,device="cuda"); k = torch.randn_like(q); v = torch.randn_like(q)
flops = 4*B*H*D*N*N
def t(fn, reps):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/reps
res = {}
ms = t(lambda: hip.forward(q,k,v,None,False,D**-0.5), 5); res["ffpa_ms"]=ms; res["ffpa_tflops"]=flops/ms/1e9
ms = t(lambda: hip.forward(q,k,v,None,False,D**-0.5,flags=hip.FLAG_NO_XCD_REMAP), 5); res["ffpa_noxcd_tflops"]=flops/ms/1e9
try:
    ms = t(lambda: torch.nn.functional.scaled_dot_product_attention(q,k,v), 3); res["sdpa_ms"]=ms; res["sdpa_tflops"]=flops/ms/1e9
    o,_ = hip.forward(q,k,v,None,False,D**-0.5); ref = torch.nn.functional.scaled_dot_product_attention(q,k,v)
    res["max_abs_vs_sdpa"] = (o.float()-ref.float()).abs().max().item()
except Exception as e:
    res["sdpa_error"] = str(e)[:200]
print("RESULT " + json.dumps(res))
"""


def run(src, timeout):
  t0 = time.time()
  try:
    p = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, timeout=timeout)
    out = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    if out:
      return json.loads(out[-1][7:]), time.time() - t0
    return {"crash": p.returncode, "stderr": p.stderr[-600:]}, time.time() - t0
  except subprocess.TimeoutExpired:
    return {"timeout": timeout}, time.time() - t0


def main():
  os.system("rocminfo 2>/dev/null | grep -E 'Marketing Name|Compute Unit|Max Clock|gfx' | head -12")
  ok = True
  for case in CASES:
    res, dt = run(CASE_SRC % {"root": ROOT, "case": case}, 240)
    print(f"CASE {case} ({dt:.1f}s): {json.dumps(res)}", flush=True)
    good = res.get("bit_equal") and res.get("fast", {}).get("max_err", 1) < 2e-2 and res.get("fast", {}).get("nan", 1) == 0
    ok = ok and bool(good)
  print("LADDER", "PASS" if ok else "FAIL", flush=True)
  for shape in ((1, 32, 8192, 512), (1, 32, 8192, 1024), (1, 32, 8192, 320)):
    res, dt = run(PERF_SRC % {"root": ROOT, "shape": shape}, 300)
    print(f"PERF {shape} ({dt:.1f}s): {json.dumps(res)}", flush=True)
  return 0 if ok else 1


if __name__ == "__main__":
  sys.exit(main())
