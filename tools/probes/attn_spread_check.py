#!/usr/bin/env python
# NOTE (round 5): this probe ran against an EXPERIMENTAL build of attention_sp_kernel (softmax spread over both stages, FW_ATTN_VAR bit 11 /
# bit 12) that was measured and not adopted -- the kernel source in the tree is the round-4 schedule, where these FW_ATTN_VAR bits select
# nothing.  Kept as the record of what profiles/r05/attn_spread_softmax_experiment_calls11_18.txt measured (docs/kernels.md).
"""Round-5 probe: the spread-softmax arm of attention_sp_kernel (FW_ATTN_VAR + 2048) against the default arm and against an fp32 softmax,
on shapes that hit every loop exit (1, 2, 3, 5, 9 tiles; ragged and exact last tiles; batches; hd 128 and 64)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fantasy_world_amd.hip_ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(3)
worst = 0.0
for hd, H in ((128, 3), (64, 5)):
    for B, Lq, Lk in ((1, 300, 64), (1, 256, 40), (1, 77, 128), (2, 515, 170), (1, 1000, 320), (1, 640, 576), (1, 2049, 1029), (2, 300, 2048), (1, 4096, 4100)):
        q = torch.randn(B * Lq, H * hd, device="cuda", generator=g).to(torch.bfloat16)
        k = torch.randn(B * Lk, H * hd, device="cuda", generator=g).to(torch.bfloat16)
        v = torch.randn(B * Lk, H * hd, device="cuda", generator=g).to(torch.bfloat16)
        qs = ops.qk_prep(q.clone(), H, hd, out_scale=ops.q_scale(hd))
        outs = {}
        for var in (192, 2240):
            ops.set_option("attn_var", var)
            outs[var] = ops.attention(qs, k, v, H, hd, batch=B, q_prescaled=True).float()
        ops.set_option("attn_var", 192)
        qf, kf, vf = (t.float().view(B, -1, H, hd).permute(0, 2, 1, 3) for t in (q, k, v))
        ref = (torch.softmax(qf @ kf.transpose(-1, -2) / math.sqrt(hd), -1) @ vf).permute(0, 2, 1, 3).reshape(B * Lq, H * hd)
        e = {var: ((o - ref).norm() / ref.norm()).item() for var, o in outs.items()}
        d = (outs[192] - outs[2240]).abs().max().item()
        nd = int((outs[192] != outs[2240]).sum())
        worst = max(worst, e[2240])
        print(f"hd {hd:3d} B {B} Lq {Lq:5d} Lk {Lk:5d}: default vs fp32 {e[192]:.3e} | spread vs fp32 {e[2240]:.3e} | max |default - spread| {d:.3e} ({nd} of {ref.numel()} elements differ)")
print("worst spread vs fp32:", worst)
