#!/bin/bash
# round 3, GPU visit 4: decode / underfilled prefill with the merge inside the launch vs the separate merge kernel
export AB_ARGS="--rounds 7 --reps 20 --case decode,decode_b8 main main:0x10000"
bash tools/gpu_round.sh "ab"
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from ffpa_attn_amd import hip
for (B,H,Hkv,Nq,Nkv,D) in ((1,4,4,512,16384,512),(1,8,8,256,32768,512),(1,2,2,512,8192,1024)):
  torch.manual_seed(0)
  q=torch.randn(B,H,Nq,D,dtype=torch.bfloat16,device='cuda'); k=torch.randn(B,Hkv,Nkv,D,dtype=torch.bfloat16,device='cuda'); v=torch.randn_like(k)
  for mil in (True, False, True, False):
    for _ in range(3): hip.forward(q,k,v,None,False,D**-0.5,merge_in_launch=mil)
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): hip.forward(q,k,v,None,False,D**-0.5,merge_in_launch=mil)
    e.record(); torch.cuda.synchronize()
    print(f"UNDERFILL B{B} H{H} Nq{Nq} Nkv{Nkv} D{D} merge_in_launch={mil}: {s.elapsed_time(e)/20*1e3:.1f} us")
PY
