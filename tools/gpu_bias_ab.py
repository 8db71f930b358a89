import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ffpa_attn_amd import hip
def t(fn, reps=10):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(reps): fn()
  e.record(); torch.cuda.synchronize()
  return s.elapsed_time(e) / reps
main = hip.load_library()
old = hip.load_library(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "ffpa_attn_amd", "variants", "libffpa_attn_hip_old.so"))
torch.manual_seed(0)
for (B,Hq,Hkv,Nq,Nkv,D) in ((1,32,32,8192,8192,512),(1,32,32,4096,4096,384),(1,32,32,8192,8192,1024)):
  q=torch.randn(B,Hq,Nq,D,dtype=torch.bfloat16,device="cuda"); k=torch.randn(B,Hkv,Nkv,D,dtype=torch.bfloat16,device="cuda"); v=torch.randn_like(k)
  m=torch.ones(Nq,Nkv,dtype=torch.bool,device="cuda").tril()
  bias=torch.zeros(1,1,Nq,Nkv,dtype=torch.bfloat16,device="cuda").masked_fill(~m,float("-inf"))
  dense=(torch.randn(1,1,Nq,Nkv,device="cuda")*0.3).to(torch.bfloat16)
  bounds=hip.mask_kv_bounds(bias,Nq,Nkv)
  res={}
  for name,lib in (("main",main),("old",old),("main2",main),("old2",old)):
    hip._lib=lib
    res[name]=(round(t(lambda: hip.forward(q,k,v,bias,False,D**-0.5,kv_bounds=bounds,return_lse=False)),4), round(t(lambda: hip.forward(q,k,v,dense,False,D**-0.5,kv_bounds=False,return_lse=False)),4), round(t(lambda: hip.forward(q,k,v,None,False,D**-0.5,return_lse=False)),4))
  o1=None
  hip._lib=main; a=hip.forward(q,k,v,dense,False,D**-0.5,kv_bounds=False)[0]
  hip._lib=old; b=hip.forward(q,k,v,dense,False,D**-0.5,kv_bounds=False)[0]
  print("BIASAB "+json.dumps({"D":D,"N":Nq,"(tril clipped, dense bias, no bias) ms":res,"equal":bool(torch.equal(a,b))}))
