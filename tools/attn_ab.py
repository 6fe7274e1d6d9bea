#!/usr/bin/env python
"""Interleaved A/B of attention kernel variants (FW_ATTN_VAR values) in ONE process on the DiT self-attention shape (and, with --all, the
bicross / VGGT shapes): alternating rounds, medians; also checks that the variants return identical bits."""
import argparse, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fantasy_world_amd.hip_ops import HipOps
ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="192,1216")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--all", action="store_true")
args = ap.parse_args()
vs = [int(v) for v in args.variants.split(",")]
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
L, L2 = 32760, 32865
shapes = [(40, 128, L, L, "dit self")] + ([(12, 96, L, L2, "bicross"), (16, 64, L2, L2, "vggt global")] if args.all else [])
for (H, hd, Lq, Lk, tag) in shapes:
    q = (torch.randn(Lq, H * hd, device="cuda", generator=g) * ops.q_scale(hd)).to(torch.bfloat16)
    k = torch.randn(Lk, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    v = torch.randn(Lk, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    vp = ops.prepare_v(v, H, hd, 1)
    outs, times = {}, {x: [] for x in vs}
    for r in range(args.rounds):
        for x in vs:
            ops.set_option("attn_var", x)
            o = torch.empty(Lq, H * hd, dtype=torch.bfloat16, device="cuda")
            fn = lambda: ops.attention(q, k, None, H, hd, out=o, v_prepared=vp, q_prescaled=True)
            fn(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.iters): fn()
            b.record(); torch.cuda.synchronize()
            times[x].append(a.elapsed_time(b) / args.iters)
            outs[x] = o
    line = f"{tag:12s} H={H} hd={hd} Lq={Lq} Lk={Lk}: "
    for x in vs:
        med = statistics.median(times[x])
        line += f" var {x}: {med:.3f} ms = {4.0*Lq*Lk*H*hd/med/1e9:7.1f} TF/s (min {min(times[x]):.3f}) |"
    line += "  identical bits: " + str(all(torch.equal(outs[vs[0]], outs[x]) for x in vs[1:]))
    line += "  rel-L2 vs the first: " + ", ".join(f"{((outs[x].float() - outs[vs[0]].float()).norm() / outs[vs[0]].float().norm()).item():.2e}" for x in vs[1:])
    print(line, flush=True)
ops.set_option("attn_var", 192)
