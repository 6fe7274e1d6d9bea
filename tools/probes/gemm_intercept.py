"""Per-tile fixed cost of the ping-pong GEMM (prologue + epilogue + work-group turnover), non-perturbing: time the same [M, N] at
several K and fit t = a + b * slabs per ROUND of 256 tiles.  M = 32760, N = 5120: exactly 10 rounds."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fantasy_world_amd.hip_ops import HipOps, Linear
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
M, N = 32760, 5120
for tag, res in (("plain bf16 out", False), ("gate + fp32 residual in place", True)):
    pts = []
    for K in (1536, 2048, 2560, 3584, 5120, 7680):
        x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        lin = Linear(torch.randn(N, K, device="cuda", generator=g).to(torch.bfloat16) * K ** -0.5, torch.zeros(N, device="cuda"))
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        xs = torch.randn(M, N, device="cuda") if res else None
        gate = torch.randn(N, device="cuda") if res else None
        fn = (lambda: ops.linear(x, lin, g1=gate, res=xs, out_f32=True, out=xs)) if res else (lambda: ops.linear(x, lin, out=out))
        ts = []
        for rep in range(5):
            fn(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(8): fn()
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 8)
        ts.sort()
        pts.append((K // 64, ts[2] * 1e3 / 10))          # us per round
    n = len(pts); sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts)
    sxx = sum(p[0] ** 2 for p in pts); sxy = sum(p[0] * p[1] for p in pts)
    b_ = (n * sxy - sx * sy) / (n * sxx - sx * sx); a_ = (sy - b_ * sx) / n
    print(f"{tag}: " + ", ".join(f"K={64*s}: {t:.1f} us/round" for s, t in pts))
    print(f"    fit: {a_:.2f} us fixed per tile + {b_:.3f} us per slab ({2*256*256*64/b_/1e6*256/1e6:.0f} TF/s asymptotic)")
