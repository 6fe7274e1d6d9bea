#!/usr/bin/env python
"""Interleaved A/B of the big-tile GEMM kernels (FW_GEMM_KERNEL values) on the DiT shapes: the variants alternate inside one
process, several rounds each, and the MEDIAN per variant is reported -- box-to-box and minute-to-minute clock drift (+-3 %) is larger
than the differences being measured, so sequential single runs cannot rank them."""
import argparse
import os
import statistics
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fantasy_world_amd.hip_ops import HipOps, Linear

ap = argparse.ArgumentParser()
ap.add_argument("--kernels", default="4,5")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--short-k", action="store_true", help="the K <= 2048 shapes of the VGGT / bicross / adapter GEMMs instead of the DiT ones")
ap.add_argument("--square", action="store_true", help="the programming guide's calibration shapes 4096^3 / 8192^3 (+ the DiT qkv shape)")
ap.add_argument("--uniform", action="store_true", help="uniform [-1, 1) operands (the guide's random fill) instead of normal")
ap.add_argument("--fp8", action="store_true", help="the fp8 linear (fw_gemm_fp8 on pre-quantised rows): kernel 9 = two-slot (round 6), 4 = four-slot")
ap.add_argument("--zeros", action="store_true", help="zero-filled operands: the same instruction stream at a fraction of the switching power (DVFS check)")
args = ap.parse_args()
# "blas" = the vendor library behind torch.matmul on the SAME operands in the SAME interleaved rounds (a yardstick of what the box
# sustains on this data; never linked or called by the product); plain shapes only (no fused residual epilogue)
kernels = [("blas", 0) if k == "blas" else tuple(int(v) for v in (k.split(":") + ["0"])[:2]) for k in args.kernels.split(",")]
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
L = 32760
SHAPES = [(L, 15360, 5120, "qkv", False), (L, 13824, 5120, "ffn0", False), (L, 5120, 5120, "o", False),
          (L, 5120, 5120, "o+res", True), (L, 5120, 13824, "ffn2+res", True)]
if args.short_k:
    SHAPES = [(32865, 3072, 1024, "vggt qkv", False), (32865, 4096, 1024, "vggt fc1", False), (32865, 1024, 1024, "vggt proj+res", True),
              (L, 5120, 1152, "bicross out1+res", True), (32865, 1024, 1152, "bicross out2+res", True), (L, 2048, 2048, "adapter g1", False)]
if args.square:
    SHAPES = [(4096, 4096, 4096, "sq4096", False), (8192, 8192, 8192, "sq8192", False), (L, 15360, 5120, "qkv", False)]
rnd = (lambda *s: torch.rand(*s, device="cuda", generator=g) * 2 - 1) if args.uniform else (lambda *s: torch.randn(*s, device="cuda", generator=g))
for (M, N, K, tag, res) in SHAPES:
    x = rnd(M, K).to(torch.bfloat16)
    lin = Linear(rnd(N, K).to(torch.bfloat16) * (1.0 if args.uniform else K ** -0.5), torch.zeros(N, device="cuda"))
    if args.zeros:
        x.zero_()
        lin.w.zero_()
    if args.fp8:
        lin = ops.pack_linear(lin.w.float(), lin.b, fp8=True)
        x = ops.quantize_fp8_rows(x)            # (e4m3 rows, scale): the GEMM alone is timed
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    xs = torch.randn(M, N, device="cuda") if res else None
    gate = torch.randn(N, device="cuda") if res else None
    times = {k: [] for k in kernels}
    for r in range(args.rounds):
        for k in kernels:
            if k[0] == "blas":
                if res:
                    times[k].append(float("nan"))
                    continue
                wt = lin.w[:N, :K].t()
                fn = lambda: torch.matmul(x, wt, out=out)
            else:
                ops.set_option("gemm_kernel", k[0])
                ops.set_option("gemm_var", k[1])
                fn = (lambda: ops.linear(x, lin, g1=gate, res=xs, out_f32=True, out=xs)) if res else (lambda: ops.linear(x, lin, out=out))
            fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            times[k].append(a.elapsed_time(b) / args.iters)
    line = f"{tag:9s} M={M} N={N} K={K}: "
    for k in kernels:
        med = statistics.median(times[k])
        line += f" kernel {k}: {med:.3f} ms = {2.0*M*N*K/med/1e9:7.1f} TF/s (min {min(times[k]):.3f})  |"
    print(line, flush=True)
ops.set_option("gemm_kernel", 9)
ops.set_option("gemm_var", 0)
