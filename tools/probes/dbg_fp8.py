import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fantasy_world_amd.hip_ops import HipOps
from oracle.ref_ops import TorchRefOps
ops, ref = HipOps("cuda:0"), TorchRefOps()
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).float()
for (M, N, K) in [(5, 64, 128), (64, 64, 64), (300, 200, 256)]:
    x, w, b = rnd(M, K, seed=3, scale=4.0), rnd(N, K, seed=4, scale=K ** -0.5), rnd(N, seed=5, scale=0.1)
    want = ref.linear_fp8(x, ref.pack_linear_fp8(w, b), out_f32=True)
    got = ops.linear_fp8(x.to(torch.bfloat16).cuda(), ops.pack_linear_fp8(w, b), out_f32=True).cpu()
    d = (got - want)
    print((M, N, K), 'rel', (d.norm() / want.norm()).item(), 'max abs', d.abs().max().item(), 'want max', want.abs().max().item())
    # double-precision reference of the same quantised operands
    q, s = ref.quantize_fp8_rows(x)
    wq = ref.pack_linear_fp8(w, b).w
    exact = (q.double() @ wq.double().t()) * s.double()[:, None] + b.to(torch.bfloat16).double()
    print('   vs fp64: gpu', ((got.double() - exact).norm() / exact.norm()).item(), ' cpu fp32', ((want.double() - exact).norm() / exact.norm()).item())
    bad = (d.abs() > 1e-3 * want.abs().max()).nonzero()
    print('   bad elements', bad.shape[0], bad[:6].tolist())
