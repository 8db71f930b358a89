import sys, torch
sys.path.insert(0, "/root/repo")
from ffpa_attn_amd import hip
def timeit(fn, reps=40, warm=100):
  for _ in range(warm): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps
torch.manual_seed(0)
B,Hq,Hkv,Nq,Nkv,D=2,32,8,8192,2048,320
q=torch.randn(B,Hq,Nq,D,dtype=torch.bfloat16,device="cuda"); k=torch.randn(B,Hkv,Nkv,D,dtype=torch.bfloat16,device="cuda"); v=torch.randn(B,Hkv,Nkv,D,dtype=torch.bfloat16,device="cuda")
pairs=sum(min(Nkv,r+1) for r in range(Nq)); fl=4*B*Hq*D*pairs
arms={"wide (default)":0, "m16 dense order":hip.FLAG_NO_WIDE_TILE|hip.FLAG_NO_HEAD_CHUNKS, "m16 head chunks":hip.FLAG_NO_WIDE_TILE}
for name,fl_ in arms.items():
  plan={}
  hip.forward(q,k,v,None,True,D**-0.5,causal_offset=0,flags=fl_,plan_out=plan,return_lse=False)
  print("ARM",name,plan.get("kernel"))
res={n:[] for n in arms}
for _ in range(4):
  for n,f in arms.items():
    res[n].append(timeit(lambda: hip.forward(q,k,v,None,True,D**-0.5,causal_offset=0,flags=f,return_lse=False)))
for n,ts in res.items():
  t=sorted(ts)[1]
  print(f"CFG4ORDER {n:18s} {t*1e3:8.1f} us {fl/t/1e9:7.1f} TF", flush=True)
