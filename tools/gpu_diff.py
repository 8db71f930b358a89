"""Where do two library builds differ?  (developer tool)   python tools/gpu_diff.py TAG_A TAG_B --shape B,H,Nq,Nkv,D [--causal]
Prints max |O_a - O_b| per 32-row block and per 128-column block, and both against fp32 math."""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ffpa_attn_amd import hip
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_ab import lib_for

ap = argparse.ArgumentParser()
ap.add_argument("a"); ap.add_argument("b")
ap.add_argument("--shape", action="append", default=[])
ap.add_argument("--causal", action="store_true")
ap.add_argument("--splits", type=int, default=0, help="num_splits for both arms (0 = the library's rule, 1 = never split)")
args = ap.parse_args()
la, lb = lib_for(args.a), lib_for(args.b)
for sh in args.shape or ["1,1,64,64,1024"]:
  B, H, Nq, Nkv, D = (int(x) for x in sh.split(","))
  torch.manual_seed(0)
  q = torch.randn(B, H, Nq, D, dtype=torch.bfloat16, device="cuda")
  k = torch.randn(B, H, Nkv, D, dtype=torch.bfloat16, device="cuda")
  v = torch.randn(B, H, Nkv, D, dtype=torch.bfloat16, device="cuda")
  outs = []
  for lib in (la, lb):
    hip._lib = lib
    plan = {}
    o, lse = hip.forward(q, k, v, None, args.causal, D ** -0.5, num_splits=args.splits, plan_out=plan)
    print("  plan", plan)
    torch.cuda.synchronize()
    outs.append((o.float(), lse))
  s = (q.float() @ k.float().transpose(-1, -2)) * D ** -0.5
  if args.causal:
    s = s.masked_fill(torch.ones(Nq, Nkv, dtype=torch.bool, device="cuda").triu(Nkv - Nq + 1), float("-inf"))
  ref = torch.softmax(s, -1) @ v.float()
  d = (outs[0][0] - outs[1][0]).abs()
  print(f"DIFF {sh} causal={args.causal}: max|a-b| {d.max().item():.3e}  a vs math {(outs[0][0] - ref).abs().max().item():.3e}  b vs math {(outs[1][0] - ref).abs().max().item():.3e}  lse a-b {(outs[0][1] - outs[1][1]).abs().max().item():.3e}")
  dl = (outs[0][1] - outs[1][1]).abs()
  print(f"  LSE elements that differ: {(dl > 0).sum().item()} of {dl.numel()}; O elements that differ: {(d > 0).sum().item()} of {d.numel()}; first rows with LSE diff: {torch.nonzero(dl.view(-1) > 0).view(-1)[:12].tolist()}")
  rows = d.amax(dim=(0, 1, 3))
  rb = rows.view(-1, min(32, Nq)).amax(1) if Nq % 32 == 0 else rows
  print("  per 32-row block:", " ".join(f"{x:.1e}" for x in rb[:16].tolist()))
  cols = d.amax(dim=(0, 1, 2)).view(-1, 128).amax(1)
  print("  per 128-col block:", " ".join(f"{x:.1e}" for x in cols.tolist()))
  e = (outs[1][0] - ref).abs().amax(dim=(0, 1, 3))
  print("  b vs math per 32-row block:", " ".join(f"{x:.1e}" for x in (e.view(-1, 32).amax(1) if Nq % 32 == 0 else e)[:16].tolist()))
