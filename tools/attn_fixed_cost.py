#!/usr/bin/env python
"""What does one attention launch cost BESIDE its key tiles?  Times the production kernel of a head size on L query rows against
Lk = 64 .. 4096 keys and fits  t(Lk) = a + b * tiles  (b = per 64-key tile; a = what a launch costs beside its tiles -- which turns out to be
HBM time, not latency: q read + O written = 670 MB, + 335 MB when the launch accumulates into O).  Cross-attention (512 + 257 keys,
two launches per DiT block, the second accumulating) lives half in `a`.  Usage (on the box): python tools/attn_fixed_cost.py [--hd 128 --heads 40]"""
import argparse, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fantasy_world_amd.hip_ops import HipOps
ap = argparse.ArgumentParser()
ap.add_argument("--hd", type=int, default=128)
ap.add_argument("--heads", type=int, default=40)
ap.add_argument("--L", type=int, default=32760)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--keys", default="64,128,256,257,512,1024,2048,4096")
args = ap.parse_args()
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
H, hd, L = args.heads, args.hd, args.L
q = (torch.randn(L, H * hd, device="cuda", generator=g) * ops.q_scale(hd)).to(torch.bfloat16)
o = torch.empty(L, H * hd, dtype=torch.bfloat16, device="cuda")
rows = []
if True:
  for acc in (False, True):
    for Lk in [int(x) for x in args.keys.split(",")]:
          k = torch.randn(Lk, H * hd, device="cuda", generator=g).to(torch.bfloat16)
          v = torch.randn(Lk, H * hd, device="cuda", generator=g).to(torch.bfloat16)
          vp = ops.prepare_v(v, H, hd, 1)
          fn = lambda: ops.attention(q, k, None, H, hd, out=o, v_prepared=vp, q_prescaled=True, accumulate=acc)
          fn(); torch.cuda.synchronize()
          ts = []
          for _ in range(5):
              a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
              a.record()
              for _ in range(args.iters): fn()
              b.record(); torch.cuda.synchronize()
              ts.append(a.elapsed_time(b) / args.iters)
          t = statistics.median(ts)
          nt = (Lk + 63) // 64
          rows.append((acc, Lk, nt, t))
          print(f"accumulate={int(acc)} Lk={Lk:5d} tiles={nt:3d}: {t*1e3:8.1f} us  ({4.0*L*Lk*H*hd/t/1e9:7.1f} TF/s)", flush=True)
for acc in (False, True):
    pts = [(nt, t) for a, Lk, nt, t in rows if a == acc and Lk % 64 == 0]
    if len(pts) < 3: continue
    n = len(pts); sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts)
    sxx = sum(p[0] ** 2 for p in pts); sxy = sum(p[0] * p[1] for p in pts)
    b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
    wgs = ((L + 255) // 256) * H
    print(f"fit accumulate={int(acc)}: fixed {a*1e3:.1f} us per launch + {b*1e3:.2f} us per tile; {wgs} work-groups = {wgs/256:.1f} rounds of 256 CUs"
          f" -> {a*1e3/(wgs/256):.2f} us fixed per work-group round, {b*1e3/(wgs/256):.3f} us per tile per work-group")
