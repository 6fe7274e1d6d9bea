#!/usr/bin/env python
"""Fuzz of the GEMM entry point: random (M, N, K, activation, bias, gate, residual type, output type, strided A) -- the default two-slot
ping-pong kernel (FW_GEMM_KERNEL 9) against the four-slot kernel (4) and the independent four-wave kernel (5) bit for bit, and
against an fp32 matmul on sampled rows.  On the box:  python tools/probes/gemm_fuzz.py [--n 60 --seed 0]"""
import argparse, os, random, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fantasy_world_amd.hip_ops import HipOps
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=60); ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
ops = HipOps("cuda:0"); rng = random.Random(args.seed); g = torch.Generator(device="cuda").manual_seed(args.seed)
fails, worst = 0, 0.0
for it in range(args.n):
    M = rng.choice([1, 97, 255, 256, 257, 1000, 2047, 2048, 2049, 4096 + 31, 9216, 16317, 32760, 32865])
    N = rng.choice([64, 409, 1024, 1152, 1536, 2048, 3072, 4096, 5120, 13824, 15360])
    K = rng.choice([64, 128, 192, 1024, 1152, 2048, 4096, 5120, 13824])
    act = rng.choice([None, None, "relu", "gelu_tanh", "gelu_erf", "silu"])
    use_bias, use_g1, use_g0 = rng.random() < 0.8, rng.random() < 0.5, rng.random() < 0.3
    res_dt = rng.choice([None, torch.float32, torch.bfloat16]); out_f32 = rng.random() < 0.5
    strided = rng.random() < 0.3
    xa = torch.randn(M, K * (2 if strided else 1), device="cuda", generator=g).to(torch.bfloat16)
    x = xa[:, :K]
    w = torch.randn(N, K, device="cuda", generator=g) * (K ** -0.5)
    b = torch.randn(N, device="cuda", generator=g) if use_bias else None
    lin = ops.pack_linear(w, b)
    g1 = torch.randn(N, device="cuda", generator=g) if use_g1 else None
    g0 = torch.randn(N, device="cuda", generator=g) if use_g0 else None
    res = None if res_dt is None else torch.randn(M, N, device="cuda", generator=g).to(res_dt)
    outs = {}
    for kern in (9, 4, 5):
        ops.set_option("gemm_kernel", kern)
        outs[kern] = ops.linear(x, lin, act=act, g1=g1, g0=g0, res=res, out_f32=out_f32).clone()
    ops.set_option("gemm_kernel", 9)
    same = torch.equal(outs[9], outs[4]) and torch.equal(outs[9], outs[5])
    rows = torch.randperm(M, device="cuda", generator=g)[:32]
    v = x[rows].float() @ lin.w.float().T
    if b is not None: v = v + b
    if act == "relu": v = torch.relu(v)
    elif act == "gelu_tanh": v = torch.nn.functional.gelu(v, approximate="tanh")
    elif act == "gelu_erf": v = torch.nn.functional.gelu(v)
    elif act == "silu": v = torch.nn.functional.silu(v)
    if g1 is not None: v = v * g1
    if g0 is not None: v = v + g0
    if res is not None: v = v + res[rows].float()
    err = float((outs[9][rows].float() - v).norm() / v.norm().clamp_min(1e-20))
    worst = max(worst, err)
    ok = same and err < (2e-5 if out_f32 else 4e-3) and bool(torch.isfinite(outs[9].float()).all())
    fails += not ok
    print(f"{'ok  ' if ok else 'FAIL'} M={M} N={N} K={K} act={act} bias={int(use_bias)} g1={int(use_g1)} g0={int(use_g0)} res={str(res_dt).replace('torch.', '')} "
          f"out={'f32' if out_f32 else 'bf16'} strided_a={int(strided)}: kernels identical={same} rel-l2={err:.2e}", flush=True)
print(f"{args.n} cases, {fails} failures, worst rel-l2 {worst:.2e}")
sys.exit(1 if fails else 0)
