import sys, torch, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffpa_attn_amd import hip
torch.manual_seed(0)
for (B,Hq,Hkv,Nkv,D) in ((1,32,8,8192,512),(1,32,8,32768,512),(4,32,8,8192,512)):
  q=torch.randn(B,Hq,1,D,dtype=torch.bfloat16,device="cuda"); k=torch.randn(B,Hkv,Nkv,D,dtype=torch.bfloat16,device="cuda"); v=torch.randn_like(k)
  f=lambda: hip.forward(q,k,v,None,False,D**-0.5,return_lse=False)[0]
  for _ in range(3): f()
  torch.cuda.synchronize()
  s=torch.cuda.Stream()
  with torch.cuda.stream(s):
    for _ in range(3): f()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
      o=f()
  torch.cuda.synchronize()
  e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
  reps=50
  e0.record()
  for _ in range(reps): g.replay()
  e1.record(); torch.cuda.synchronize()
  tg=e0.elapsed_time(e1)/reps
  e0.record()
  for _ in range(reps): f()
  e1.record(); torch.cuda.synchronize()
  te=e0.elapsed_time(e1)/reps
  byt=2*B*Hkv*Nkv*D*2
  print("GRAPHDEC "+json.dumps({"shape":f"B{B} Hq{Hq}/Hkv{Hkv} Nq1 Nkv{Nkv} D{D}","eager_us":round(te*1e3,1),"graph_us":round(tg*1e3,1),"graph_GBps":round(byt/tg/1e6,1)}))
