# NOTE (round 5): this probe ran against an EXPERIMENTAL build of attention_sp_kernel (softmax spread over both stages, FW_ATTN_VAR bit 11 /
# bit 12) that was measured and not adopted -- the kernel source in the tree is the round-4 schedule, where these FW_ATTN_VAR bits select
# nothing.  Kept as the record of what profiles/r05/attn_spread_softmax_experiment_calls11_18.txt measured (docs/kernels.md).
import math, os, sys, torch
sys.path.insert(0, "/root/repo")
from fantasy_world_amd.hip_ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(3)
hd, H = 128, 3
for B, Lq, Lk in ((1, 256, 64), (1, 2049, 1029), (1, 4096, 4100), (1, 8192, 8192), (1, 512, 32760), (1, 32760, 32760), (1, 32760, 16384)):
    q = torch.randn(B * Lq, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    k = torch.randn(B * Lk, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    v = torch.randn(B * Lk, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    qs = ops.qk_prep(q.clone(), H, hd, out_scale=ops.q_scale(hd))
    outs = {}
    for var in (192, 1216, 2240, 3264):
        ops.set_option("attn_var", var)
        outs[var] = ops.attention(qs, k, v, H, hd, batch=B, q_prescaled=True).float()
    ops.set_option("attn_var", 192)
    names = {192: "spread/desc", 1216: "spread/ptr", 2240: "classic/desc", 3264: "classic/ptr"}
    line = f"Lq {Lq} Lk {Lk}: "
    ks = list(outs)
    for i in range(4):
        for j in range(i + 1, 4):
            n = int((outs[ks[i]] != outs[ks[j]]).sum())
            line += f"{names[ks[i]]} vs {names[ks[j]]}: {n} | "
    d = (outs[192] != outs[2240])
    rows = d.any(dim=1).nonzero().flatten()
    line += f" rows differing (spread vs classic desc): {len(rows)} of {d.shape[0]}, first {rows[:6].tolist()}"
    print(line)
