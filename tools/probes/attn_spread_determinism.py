# NOTE (round 5): this probe ran against an EXPERIMENTAL build of attention_sp_kernel (softmax spread over both stages, FW_ATTN_VAR bit 11 /
# bit 12) that was measured and not adopted -- the kernel source in the tree is the round-4 schedule, where these FW_ATTN_VAR bits select
# nothing.  Kept as the record of what profiles/r05/attn_spread_softmax_experiment_calls11_18.txt measured (docs/kernels.md).
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fantasy_world_amd.hip_ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(3)
hd, H = 128, 3
for B, Lq, Lk in ((1, 4096, 4100), (1, 32760, 16384)):
    q = torch.randn(B * Lq, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    k = torch.randn(B * Lk, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    v = torch.randn(B * Lk, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    qs = ops.qk_prep(q.clone(), H, hd, out_scale=ops.q_scale(hd))
    for var, name in ((192, "spread/desc"), (2240, "classic/desc"), (192 + 4096, "spread/desc + 32 wait states before the first early exp")):
        ops.set_option("attn_var", var)
        runs = [ops.attention(qs, k, v, H, hd, batch=B, q_prescaled=True).float() for _ in range(4)]
        torch.cuda.synchronize()
        nd = [int((runs[0] != r).sum()) for r in runs[1:]]
        md = [float((runs[0] - r).abs().max()) for r in runs[1:]]
        print(f"Lq {Lq} Lk {Lk} {name}: elements differing between run 0 and runs 1..3: {nd}, max abs diff {md}")
    ops.set_option("attn_var", 192)
