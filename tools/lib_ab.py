#!/usr/bin/env python
"""Interleaved A/B of TWO BUILDS of libfw_mi355x.so in ONE process (compiler-flag experiments: VERDICT r05 next 8, `-fno-associative-math`
for the MFMA files): the production attention launches (DiT self hd 128, bicross hd 96, VGGT global hd 64, fp8 hd 128) and the two-slot
GEMM on the qkv / ffn0 / o + gate + fp32 residual / ffn2 shapes, alternating rounds, medians, and whether the two builds return the
same bits.

    FW_BUILD_TAG=noassoc FW_MFMA_EXTRA_FLAGS=-fno-associative-math bash fantasy_world_amd/csrc/build.sh     # here, no GPU needed
    python tools/lib_ab.py --b fantasy_world_amd/libfw_mi355x.noassoc.so                                   # on the box
"""
import argparse, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fantasy_world_amd import hip_ops
from fantasy_world_amd.hip_ops import HipOps
ap = argparse.ArgumentParser()
ap.add_argument("--a", default=hip_ops.LIB_PATH)
ap.add_argument("--b", required=True)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--heads", type=int, default=8, help="heads of the hd-128 launches (8 = one fifth of a block's launch; 40 = the launch)")
args = ap.parse_args()
libs = {}
for tag, path in (("A", args.a), ("B", args.b)):
    o = HipOps("cuda:0")
    o.lib = hip_ops.load_library(os.path.abspath(path), cache=False)
    libs[tag] = o
print(f"A = {args.a}\nB = {args.b}", flush=True)
g = torch.Generator(device="cuda").manual_seed(0)
L, L2, D, Fd = 32760, 32865, 5120, 13824


def ab(tag, flops, make):
    """make(ops) -> (fn, out tensor); alternating rounds, medians."""
    fns = {t: make(o) for t, o in libs.items()}
    times = {t: [] for t in libs}
    for r in range(args.rounds):
        for t, (fn, _) in fns.items():
            fn(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.iters): fn()
            b.record(); torch.cuda.synchronize()
            times[t].append(a.elapsed_time(b) / args.iters)
    ma, mb = statistics.median(times["A"]), statistics.median(times["B"])
    same = torch.equal(fns["A"][1], fns["B"][1])
    print(f"{tag:34s} A {ma:8.3f} ms = {flops/ma/1e9:7.1f} TF/s | B {mb:8.3f} ms = {flops/mb/1e9:7.1f} TF/s | B vs A {100*(ma/mb-1):+5.2f} % | identical bits: {same}", flush=True)


for (H, hd, Lq, Lk, tag) in [(args.heads, 128, L, L, "attn hd128 dit self"), (12, 96, L, L2, "attn hd96 bicross"), (16, 64, L2, L2, "attn hd64 vggt global")]:
    ops0 = libs["A"]
    q = (torch.randn(Lq, H * hd, device="cuda", generator=g) * ops0.q_scale(hd)).to(torch.bfloat16)
    k = torch.randn(Lk, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    v = torch.randn(Lk, H * hd, device="cuda", generator=g).to(torch.bfloat16)

    def make(ops, q=q, k=k, v=v, H=H, hd=hd, Lq=Lq):
        vp = ops.prepare_v(v, H, hd, 1)
        o = torch.empty(Lq, H * hd, dtype=torch.bfloat16, device="cuda")
        return (lambda: ops.attention(q, k, None, H, hd, out=o, v_prepared=vp, q_prescaled=True)), o
    ab(tag, 4.0 * Lq * Lk * H * hd, make)
    if hd == 128:
        def make8(ops, q=q, k=k, v=v, H=H, hd=hd, Lq=Lq, Lk=Lk):
            q8 = ops.cast_fp8((q.float() * 8).to(torch.bfloat16)); k8 = ops.cast_fp8(k); vt8, _ = ops.prepare_v_fp8(v, H, hd)
            o = torch.empty(Lq, H * hd, dtype=torch.bfloat16, device="cuda")
            return (lambda: ops.attention_fp8(q8, k8, vt8, H, hd, Lk, out=o)), o
        ab("attn fp8 hd128 dit self", 4.0 * Lq * Lk * H * hd, make8)

for (M, N, K, res, tag) in [(L, 3 * D, D, False, "gemm qkv"), (L, Fd, D, False, "gemm ffn0 (gelu)"), (L, D, D, True, "gemm o + gate + fp32 residual"),
                            (L, D, Fd, True, "gemm ffn2 + gate + fp32 residual"), (L2, 3072, 1024, False, "gemm vggt qkv (K = 1024)")]:
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02)
    bias = torch.randn(N, device="cuda", generator=g)
    gate = torch.randn(N, device="cuda", generator=g)
    stream = torch.randn(M, N, device="cuda", generator=g) if res else None

    def make(ops, x=x, w=w, bias=bias, gate=gate, stream=stream, tag=tag):
        lin = ops.pack_linear(w, bias)
        if stream is not None:
            o = torch.empty_like(stream)
            return (lambda: ops.linear(x, lin, g1=gate, res=stream, out_f32=True, out=o)), o
        o = torch.empty(x.shape[0], lin.N, dtype=torch.bfloat16, device="cuda")
        return (lambda: ops.linear(x, lin, act="gelu_tanh" if "gelu" in tag else None, out=o)), o
    ab(tag, 2.0 * M * N * K, make)
    del x, w, stream
