import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fantasy_world_amd.hip_ops import HipOps, Linear
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K) in [(97, 1024, 4096), (97, 1024, 1024), (97, 4096, 1024), (97, 3072, 1024), (512, 10240, 5120), (257, 10240, 1280), (128, 5120, 5120)]:
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    lin = Linear(torch.randn(N, K, device="cuda", generator=g).bfloat16() * K ** -0.5, torch.zeros(N, device="cuda"))
    fn = lambda: ops.linear(x, lin)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    print(f"M={M} N={N} K={K}: {a.elapsed_time(b)/20*1e3:.1f} us")
