"""Calibration against the programming guide's own best HIP-source GEMM (cdna_hip_programming.md, "The 256^2 8-phase template":
~1320-1340 TF/s at 4096^3 and ~1470 at 8192^3 on uniform random [-1, 1) operands, 1563 / 1728 on zeros): the ping-pong kernel on
the same square shapes and the same fills, interleaved rounds, medians.  Also the DiT qkv shape for the fill dependence."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fantasy_world_amd.hip_ops import HipOps, Linear
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)


def fill(kind, *shape):
    if kind == "zeros":
        return torch.zeros(*shape, device="cuda", dtype=torch.bfloat16)
    if kind == "uniform":
        return (torch.rand(*shape, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    return torch.randn(*shape, device="cuda", generator=g).to(torch.bfloat16)


cases = []
for (M, N, K) in ((4096, 4096, 4096), (8192, 8192, 8192), (32760, 15360, 5120)):
    for kind in ("uniform", "normal", "zeros"):
        x = fill(kind, M, K)
        lin = Linear(fill(kind, N, K), torch.zeros(N, device="cuda"))
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        cases.append((M, N, K, kind, x, lin, out, []))
for rnd in range(5):
    for (M, N, K, kind, x, lin, out, ts) in cases:
        fn = lambda: ops.linear(x, lin, out=out)
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20 if M <= 8192 else 6
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps)
for (M, N, K, kind, x, lin, out, ts) in cases:
    ts.sort()
    print(f"M={M:6d} N={N:6d} K={K:5d} {kind:8s} median {ts[2]:8.3f} ms = {2.0*M*N*K/ts[2]/1e9:7.1f} TF/s   best {2.0*M*N*K/ts[0]/1e9:7.1f}")
