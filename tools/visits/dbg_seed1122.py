import sys, math, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import test_varlen_gpu as tv
from ffpa_attn_amd import hip
import oracle as fo_pkg
from test_fwd_gpu import fo, _f32
seed=1122
rng = np.random.default_rng(1000 + seed)
nseq = int(rng.integers(1, 13))
lens_q = [int(x) for x in rng.choice([0, 1, 7, 31, 33, 64, 127, 128, 129, 200, 333, 512, 700], size=nseq)]
lens_k = lens_q if rng.random() < 0.4 else [int(x) for x in rng.choice([0, 1, 17, 64, 65, 128, 255, 256, 300, 640], size=nseq)]
hkv = int(rng.choice([1, 2, 4])); hq = hkv * int(rng.choice([1, 2, 4])); d = int(rng.choice([128, 256, 320, 512, 640, 1024]))
dtype = torch.bfloat16 if rng.random() < 0.6 else torch.float16
causal = bool(rng.random() < 0.5)
print(lens_q, lens_k, hq, hkv, d, dtype, causal)
q, k, v = tv._make(lens_q, lens_k, hq, hkv, d, dtype, seed=seed)
out, lse = hip.varlen_forward(q, k, v, tv._cu(lens_q), tv._cu(lens_k), max(lens_q), max(lens_k), causal, 1.0 / math.sqrt(d))
n=lens_q[0]
qi, ki, vi = tv._seq(q,0,n), tv._seq(k,0,n), tv._seq(v,0,n)
qb, dt = fo.torch_to_bits(qi); kb,_=fo.torch_to_bits(ki); vb,_=fo.torch_to_bits(vi)
_, o32, l, (pmax, p2) = fo.oracle_forward(qb, kb, vb, dt, causal=True, causal_offset=0, block_keys=64, threshold=8.0, return_pmax="both")
got=_f32(tv._seq(out,0,n))
err=np.abs(got-o32)
idx=np.unravel_index(np.argmax(err), err.shape)
print('worst', idx, 'err', err[idx], 'want', o32[idx], 'got', got[idx], 'pmax row', pmax[idx[:3]], 'visible keys', idx[2]+1)
# exact math for that row
h,r=idx[1],idx[2]
qq=qi[0,h,r].float(); kk=ki[0,h//(hq//hkv),:r+1].float(); vv=vi[0,h//(hq//hkv),:r+1].float()
s=(kk@qq)/math.sqrt(d); p=torch.softmax(s,0); print('p', p.tolist()[:8], 'exact', float((p[:,None]*vv).sum(0)[idx[3]]), 'v col', vv[:, idx[3]].tolist()[:8])
pb=p.to(torch.bfloat16).float(); print('p rounded to bf16 then PV / sum p (unrounded l):', float((pb[:,None]*vv).sum(0)[idx[3]]/1.0))
