"""Round-5 probe: is the fp8 single-stream kernel deterministic run to run at production sizes (the spread-softmax experiment on the bf16
kernel was not: profiles/r05/attn_spread_softmax_experiment_calls11_18.txt)?  Every arm 4 times on the same inputs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fantasy_world_amd.hip_ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(3)
hd, H = 128, 8
REPS = int(os.environ.get('REPS', 5))
for Lq, Lk in ((4096, 4100), (32760, 32760)) + (((111600, 111600),) if os.environ.get('BIG') else ()):
    q = torch.randn(Lq, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    k = torch.randn(Lk, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    v = torch.randn(Lk, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    q8 = ops.cast_fp8(ops.qk_prep(q.clone(), H, hd, out_scale=ops.q_scale_fp8(hd)))
    k8 = ops.cast_fp8(k)
    vt8, lk = ops.prepare_v_fp8(v, H, hd)
    qs = ops.qk_prep(q.clone(), H, hd, out_scale=ops.q_scale(hd))
    for var, name in ((192, "fp8 single-stream (default)"), (11, "fp8 single-stream in phase"), (9, "fp8 ping-pong")):
        ops.set_option("attn_var", var)
        runs = [ops.attention_fp8(q8, k8, vt8, H, hd, lk) for _ in range(REPS)]
        torch.cuda.synchronize()
        print(f"Lq {Lq} Lk {Lk} {name}: runs differing from run 0: {sum(int(not torch.equal(runs[0], r)) for r in runs[1:])} of {len(runs) - 1}")
    ops.set_option("attn_var", 192)
    runs = [ops.attention(qs, k, v, H, hd, q_prescaled=True) for _ in range(REPS)]
    print(f"Lq {Lq} Lk {Lk} bf16 kernel (default): runs differing from run 0: {sum(int(not torch.equal(runs[0], r)) for r in runs[1:])} of {len(runs) - 1}")
