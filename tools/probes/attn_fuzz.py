#!/usr/bin/env python
"""Fuzz of the attention request paths: random (heads, head size, batch, Lq, Lk, layout) -- the default kernels (tile requests by SGPR
descriptor) against FW_ATTN_VAR bit 10 (pointer form) bit for bit, and against an fp32 softmax on sampled rows.  On the box:
    python tools/probes/attn_fuzz.py [--n 80 --seed 0]"""
import argparse, math, os, random, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fantasy_world_amd.hip_ops import HipOps
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=80); ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
ops = HipOps("cuda:0"); rng = random.Random(args.seed); g = torch.Generator(device="cuda").manual_seed(args.seed)
worst, fails = 0.0, 0
for it in range(args.n):
    hd = rng.choice([64, 96, 128]); heads = rng.randint(1, 6); batch = rng.choice([1, 1, 2, 3])
    Lq = rng.choice([1, 31, 64, 255, 256, 257, 300, 777, 1030]); Lk = rng.choice([1, 2, 31, 63, 64, 65, 127, 128, 129, 257, 512, 1000, 1565, 2049])
    fused = rng.random() < 0.5; split = rng.random() < 0.5
    D = heads * hd
    q = (torch.randn(batch * Lq, D, device="cuda", generator=g) * ops.q_scale(hd) * rng.choice([1.0, 1.0, 4.0])).to(torch.bfloat16)
    if fused:
        kv = torch.randn(batch * Lk, 2 * D, device="cuda", generator=g).to(torch.bfloat16); k, v = kv[:, :D], kv[:, D:]
    else:
        k = torch.randn(batch * Lk, D, device="cuda", generator=g).to(torch.bfloat16); v = torch.randn(batch * Lk, D, device="cuda", generator=g).to(torch.bfloat16)
    ops.split_kv = split
    outs = []
    for var in (192, 192 + 1024):
        ops.set_option("attn_var", var)
        outs.append(ops.attention(q, k, v, heads, hd, batch=batch, q_prescaled=True).clone())
    ops.set_option("attn_var", 192)
    same = torch.equal(outs[0], outs[1])
    # fp32 reference on up to 64 sampled query rows of batch item 0, head 0
    rows = torch.randperm(Lq, device="cuda", generator=g)[:64]
    s = (q[rows, :hd].float() @ k[:Lk, :hd].float().T) * 0.6931471805599453
    want = torch.softmax(s, -1) @ v[:Lk, :hd].float()
    err = float((outs[0][rows, :hd].float() - want).norm() / want.norm().clamp_min(1e-20))
    worst = max(worst, err)
    ok = same and err < 8e-3 and bool(torch.isfinite(outs[0].float()).all())
    fails += not ok
    print(f"{'ok  ' if ok else 'FAIL'} hd={hd} heads={heads} batch={batch} Lq={Lq} Lk={Lk} fused_kv={int(fused)} split_kv={int(split)}: identical={same} rel-l2={err:.2e}", flush=True)
print(f"{args.n} cases, {fails} failures, worst rel-l2 {worst:.2e}")
sys.exit(1 if fails else 0)
