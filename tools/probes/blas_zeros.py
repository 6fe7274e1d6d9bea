"""Vendor BLAS (torch.matmul) on the qkv shape with random vs zero-filled operands: is the yardstick kernel clock / power limited too?"""
import torch
g = torch.Generator(device="cuda").manual_seed(0)
M, N, K = 32760, 15360, 5120
for zeros in (False, True, False, True):
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = torch.randn(N, K, device="cuda", generator=g).bfloat16()
    if zeros:
        x.zero_(); w.zero_()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(x, w.t(), out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        torch.matmul(x, w.t(), out=out)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"vendor BLAS qkv {'zeros ' if zeros else 'random'}: {ms:.3f} ms = {2.0*M*N*K/ms/1e9:.0f} TF/s", flush=True)
