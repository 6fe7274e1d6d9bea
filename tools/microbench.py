#!/usr/bin/env python
"""Per-kernel timings on one MI355X (HIP events on the launch stream): the GEMM and attention shapes of the
81x480x832 forward.  Prints one line per shape: ms, TFLOP/s (algorithmic), fraction of the 2.5 PF bf16 MFMA peak;
HBM-bound glue kernels report GB/s against 8 TB/s."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fantasy_world_amd.hip_ops import HipOps, Linear  # noqa: E402


def timeit(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--L", type=int, default=32760)
    ap.add_argument("--gemm-variants", default="", help="comma list of kernel:var pairs to A/B, e.g. 1:0,1:1,0:0")
    ap.add_argument("--attn-variants", default="", help="comma list of FW_ATTN_VAR values to A/B")
    ap.add_argument("--blas-ceiling", action="store_true",
                    help="also time torch.matmul (hipBLASLt / rocBLAS) on the big GEMM shapes: what the vendor library reaches on "
                         "THIS box -- a yardstick for the hand-written kernels, never part of the product")
    args = ap.parse_args()
    ops = HipOps("cuda:0")
    dev = "cuda:0"
    L = args.L
    L2 = L + 105
    res = []
    g = torch.Generator(device=dev).manual_seed(0)
    rb = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)

    gvars = [tuple(int(a) for a in v.split(":")) for v in args.gemm_variants.split(",") if v] or [(9, 0)]
    avars = [int(v) for v in args.attn_variants.split(",") if v] or [192]
    if args.only in ("", "gemm", "gemmonly"):
      for (gk, gv) in gvars:
        ops.set_option("gemm_kernel", gk)
        ops.set_option("gemm_var", gv)
        print(f"== gemm kernel={gk} var={gv}", flush=True)
        for (M, N, K, tag) in [(L, 15360, 5120, "dit qkv"), (L, 5120, 5120, "dit o/q"), (L, 13824, 5120, "ffn0"),
                               (L, 5120, 13824, "ffn2"), (L2, 3072, 1024, "vggt qkv"), (L2, 4096, 1024, "vggt fc1"),
                               (L2, 1024, 4096, "vggt fc2"), (L, 2304, 5120, "bicross qv1"), (L, 5120, 1152, "bicross out1"),
                               (L, 1024, 5120, "adapter g20"), (L, 2048, 2048, "adapter g1"), (512, 10240, 5120, "ctx kv")]:
            x = rb(M, K)
            lin = Linear(rb(N, K) * (K ** -0.5), torch.zeros(N, device=dev))
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            ms = timeit(lambda: ops.linear(x, lin, out=out), args.iters)
            tf = 2.0 * M * N * K / ms / 1e9
            res.append(dict(kernel="gemm", variant=f"{gk}:{gv}", tag=tag, M=M, N=N, K=K, ms=ms, tflops=tf, frac=tf / 2500))
            print(f"gemm {tag:14s} M={M:6d} N={N:6d} K={K:6d}  {ms:8.3f} ms  {tf:7.1f} TF/s  {tf/25:5.1f}% of peak", flush=True)
            if tag in ("dit o/q", "vggt fc2"):
                # the same GEMM as the engine runs it: gate * (acc + bias) + fp32 residual, in place on the fp32 stream
                xs = torch.randn(M, N, device=dev)
                gate = torch.randn(N, device=dev)
                ms = timeit(lambda: ops.linear(x, lin, g1=gate, res=xs, out_f32=True, out=xs), args.iters)
                tf = 2.0 * M * N * K / ms / 1e9
                res.append(dict(kernel="gemm+res_f32", variant=f"{gk}:{gv}", tag=tag, M=M, N=N, K=K, ms=ms, tflops=tf))
                print(f"gemm {tag:14s} + gate + fp32 residual in place      {ms:8.3f} ms  {tf:7.1f} TF/s", flush=True)
                del xs, gate
            if args.blas_ceiling and (gk, gv) == gvars[0] and M >= 2048 and N >= 1024:
                wt = lin.w.t()
                ms = timeit(lambda: torch.matmul(x, wt, out=out), args.iters)
                tf = 2.0 * M * N * K / ms / 1e9
                res.append(dict(kernel="vendor_blas", tag=tag, M=M, N=N, K=K, ms=ms, tflops=tf))
                print(f"     vendor BLAS (torch.matmul) same shape          {ms:8.3f} ms  {tf:7.1f} TF/s", flush=True)
            del x, lin, out
      ops.set_option("gemm_kernel", 9)
      ops.set_option("gemm_var", 0)
    if args.only in ("", "fp8", "gemm"):
        print("== fp8 linear (quantise rows + 256x256x128 ping-pong GEMM on the scaled MFMA)", flush=True)
        for (M, N, K, tag) in [(L, 15360, 5120, "dit qkv"), (L, 5120, 5120, "dit o/q"), (L, 13824, 5120, "ffn0"),
                               (L, 5120, 13824, "ffn2")]:
            x = rb(M, K)
            lin = ops.pack_linear_fp8(rb(N, K) * (K ** -0.5), torch.zeros(N, device=dev))
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            xq = ops.quantize_fp8_rows(x)
            ms_q = timeit(lambda: ops.quantize_fp8_rows(x), args.iters)
            ms_g = timeit(lambda: ops.linear(xq, lin, out=out), args.iters)
            ms = timeit(lambda: ops.linear(x, lin, out=out), args.iters)
            tf, tfg = 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms_g / 1e9
            res.append(dict(kernel="gemm_fp8", tag=tag, M=M, N=N, K=K, ms=ms, ms_gemm_only=ms_g, ms_quant=ms_q, tflops=tf,
                            tflops_gemm_only=tfg, frac_fp8_peak=tfg / 5000))
            print(f"fp8  {tag:14s} M={M:6d} N={N:6d} K={K:6d}  GEMM {ms_g:7.3f} ms {tfg:7.1f} TF/s ({tfg/50:4.1f}% of the fp8 peak)  "
                  f"quant {ms_q:6.3f} ms ({M*K*3/ms_q/1e6:6.0f} GB/s)  linear {ms:7.3f} ms {tf:7.1f} TF/s", flush=True)
            if tag in ("dit o/q", "ffn2"):
                xs = torch.randn(M, N, device=dev)
                gate = torch.randn(N, device=dev)
                ms_g = timeit(lambda: ops.linear(xq, lin, g1=gate, res=xs, out_f32=True, out=xs), args.iters)
                print(f"fp8  {tag:14s} GEMM + gate + fp32 residual in place   {ms_g:7.3f} ms {2.0*M*N*K/ms_g/1e9:7.1f} TF/s", flush=True)
                del xs, gate
            del x, lin, out, xq
    if args.only in ("", "attn"):
      for av in avars:
        ops.set_option("attn_var", av)
        print(f"== attention var={av}", flush=True)
        for (H, hd, B, Lq, Lk, tag) in [(40, 128, 1, L, L, "dit self"), (40, 128, 1, L, 512, "dit cross txt"),
                                        (12, 96, 1, L, L2, "bicross"), (12, 96, 1, L2, L, "bicross dir2"),
                                        (16, 64, 1, L2, L2, "vggt global"),
                                        (16, 64, 21, L2 // 21, L2 // 21, "vggt frame")]:
            q, k, v = rb(B * Lq, H * hd), rb(B * Lk, H * hd), rb(B * Lk, H * hd)
            if av >= 64:
                q = (q.float() * ops.q_scale(hd)).to(torch.bfloat16)     # what qk_prep(out_scale=...) hands the fast kernel
            vp = ops.prepare_v(v, H, hd, B)
            out = torch.empty(B * Lq, H * hd, dtype=torch.bfloat16, device=dev)
            ms = timeit(lambda: ops.attention(q, k, None, H, hd, batch=B, out=out, v_prepared=vp, q_prescaled=av >= 64), args.iters)
            tf = 4.0 * B * Lq * Lk * H * hd / ms / 1e9
            res.append(dict(kernel="attention", variant=av, tag=tag, H=H, hd=hd, B=B, Lq=Lq, Lk=Lk, ms=ms, tflops=tf, frac=tf / 2500))
            print(f"attn {tag:14s} H={H:3d} hd={hd:3d} B={B:2d} Lq={Lq:6d} Lk={Lk:6d}  {ms:8.3f} ms  {tf:7.1f} TF/s  {tf/25:5.1f}% of peak", flush=True)
            ms = timeit(lambda: ops.prepare_v(v, H, hd, B), args.iters)
            gb = 2.0 * v.numel() * 2 / ms / 1e6
            print(f"     v_transpose {tag:14s} {ms:8.3f} ms  {gb:7.1f} GB/s", flush=True)
            del q, k, v, vp, out
      ops.set_option("attn_var", 192)
    if args.only in ("", "glue"):
        x = torch.randn(L, 5120, device=dev)
        sc = torch.randn(5120, device=dev)
        ms = timeit(lambda: ops.layernorm(x, scale=sc, shift=sc), args.iters)
        print(f"layernorm_mod [L,5120] f32->bf16 {ms:8.3f} ms  {L*5120*6/ms/1e6:7.1f} GB/s", flush=True)
        res.append(dict(kernel="layernorm_mod", ms=ms, gbps=L * 5120 * 6 / ms / 1e6))
        from fantasy_world_amd import rope
        tab = rope.rope3d_table(128, 21, 30, 52)[:L].to(dev) if L <= 32760 else None
        if tab is not None:
            qkv = rb(L, 15360)
            nw = torch.ones(5120, device=dev)
            ms = timeit(lambda: ops.qk_prep(qkv[:, :5120], 40, 128, "rms_full", nw, None, 1e-6, "interleaved", tab), args.iters)
            print(f"qk_prep rms+rope3d [L,5120] in place {ms:8.3f} ms  {L*5120*4/ms/1e6:7.1f} GB/s", flush=True)
            res.append(dict(kernel="qk_prep", ms=ms, gbps=L * 5120 * 4 / ms / 1e6))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "microbench.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
