import sys, torch
sys.path.insert(0, '.')
import importlib
hip_ops = importlib.import_module('fantasy_world_amd.hip_ops')
ops = hip_ops.HipOps('cuda:0')
torch.manual_seed(0)
def run(heads, hd, Lq, Lk):
    q = (torch.randn(Lq, heads*hd, device='cuda') * ops.q_scale(hd)).to(torch.bfloat16)
    k = torch.randn(Lk, heads*hd, device='cuda').to(torch.bfloat16)
    v = torch.randn(Lk, heads*hd, device='cuda').to(torch.bfloat16)
    outs = {}
    for var in (129, 131):
        ops.set_option('attn_var', var)
        outs[var] = ops.attention(q, k, v, heads, hd, q_prescaled=True).float()
    d = (outs[131] - outs[129]).abs()
    print(f'H{heads} hd{hd} Lq{Lq} Lk{Lk}: max err {d.max().item():.4g}  nan {torch.isnan(outs[131]).sum().item()}')
    bad_rows = (d.max(dim=1).values > 0.02).nonzero().flatten().tolist()
    print('  bad rows', len(bad_rows), bad_rows[:40])
    for h in range(heads):
        bc = (d[:, h*hd:(h+1)*hd].max(dim=0).values > 0.02).nonzero().flatten().tolist()
        print('  head', h, 'bad cols', len(bc), bc[:40])
for c in [(1,128,64,64),(1,128,64,32),(1,128,256,64),(1,128,64,128),(1,128,64,640),(1,96,64,64),(1,64,64,64),(1,128,256,2048)]:
    run(*c)
