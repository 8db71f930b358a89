"""How long does a burst of launches take to reach its steady clock? (developer tool)  For a workload: idle the GPU, then N launches back to
back with one HIP event pair per launch and the hwmon shader clock / socket power sampled by a thread; prints the per-launch time of launches
0 .. N in groups, next to the clock samples that fall into each group.  Answers whether `--warmup 5 --steps 20` of a 0.5 ms kernel is timed on
a clock that a longer burst would not see (profiles/r04_clock_ramp.txt)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (DeviceTelemetry)
from ffpa_attn_amd import hip  # noqa: E402

CASES = {"cross": (1, 32, 32, 1024, 8192, 512), "cfg2": (1, 32, 32, 8192, 8192, 512), "cfg3": (1, 32, 32, 8192, 8192, 1024), "cfg4_nomask": (2, 32, 8, 8192, 2048, 320)}
N = int(os.environ.get("N", "400"))
hip.load_library()
tel = bench.DeviceTelemetry(0)
for name in os.environ.get("ONLY", "cross,cfg2").split(","):
  B, Hq, Hkv, Nq, Nkv, D = CASES[name]
  torch.manual_seed(0)
  q = torch.randn(B, Hq, Nq, D, dtype=torch.bfloat16, device="cuda")
  k = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  v = torch.randn(B, Hkv, Nkv, D, dtype=torch.bfloat16, device="cuda")
  flops = 4 * B * Hq * D * Nq * Nkv
  hip.forward(q, k, v, None, False, D ** -0.5, return_lse=False)
  torch.cuda.synchronize()
  for idle_s in (2.0, 0.0):
    n = N if Nq * Nkv * D < 3e10 else N // 4
    time.sleep(idle_s)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    tel.start(0.0005)
    t0 = time.perf_counter()
    for a, b in ev:
      a.record()
      hip.forward(q, k, v, None, False, D ** -0.5, return_lse=False)
      b.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    tel.stop()
    ms = [a.elapsed_time(b) for a, b in ev]
    start = [ev[0][0].elapsed_time(a) for a, _ in ev]
    samples = list(tel.samples)  # (mhz, watts) every 0.5 ms, evenly over the burst (the host runs ahead of the GPU only by the queue depth)
    groups = [(0, 5), (5, 25), (25, 50), (50, 100), (100, 200), (200, 400)]
    print(f"RAMP {name} after {idle_s:.0f} s idle: {n} launches in {wall * 1e3:.1f} ms wall, {len(samples)} clock samples")
    for lo, hi in groups:
      if lo >= n:
        break
      hi = min(hi, n)
      seg = sorted(ms[lo:hi])
      t_lo, t_hi = start[lo], start[hi - 1] + ms[hi - 1]
      ss = [s for i, s in enumerate(samples) if t_lo <= (i + 0.5) * wall * 1e3 / max(len(samples), 1) <= t_hi]
      clk = f"{sum(s[0] for s in ss) / len(ss):6.0f} MHz {sum(s[1] for s in ss) / len(ss):5.0f} W ({len(ss)} samples)" if ss else "no sample"
      print(f"RAMP   launches {lo:3d}..{hi - 1:3d} (t = {t_lo:7.1f} .. {t_hi:7.1f} ms): median {seg[len(seg) // 2]:.4f} ms  min {seg[0]:.4f}  max {seg[-1]:.4f}  "
            f"{flops / seg[len(seg) // 2] / 1e9:7.1f} TF | {clk}", flush=True)
