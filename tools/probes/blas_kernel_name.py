"""Which kernel does the vendor BLAS (torch.matmul -> hipBLASLt / rocBLAS) run for the big GEMM shapes?  Run under
rocprofv3 --kernel-trace --stats; the Tensile kernel name encodes macro tile, MFMA shape, wave layout, LDS options."""
import torch
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K) in [(32760, 5120, 5120), (32760, 15360, 5120), (32760, 5120, 13824)]:
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = torch.randn(N, K, device="cuda", generator=g).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        torch.matmul(x, w.t(), out=out)
    torch.cuda.synchronize()
