#!/usr/bin/env python
"""ONE kind of launch per kernel, for the PMC traffic passes (profiles/r03/pmc_traffic.json): the DiT self-attention launch of the
headline workload (L = 32760, 40 heads x 128) ALONE, so its FETCH_SIZE / WRITE_SIZE averages need no subtraction of an estimated
cross-attention share, plus the [L,5120] fp32 -> bf16 LayerNorm launch whose bytes are known exactly
(calibration of the gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section).
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o g -- python tools/attn_self_only.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fantasy_world_amd.hip_ops import HipOps   # noqa: E402

L, H, hd, iters = 32760, 40, 128, int(os.environ.get("ITERS", "3"))
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(L, 3 * H * hd, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
q, k, v = qkv[:, :H * hd], qkv[:, H * hd:2 * H * hd], qkv[:, 2 * H * hd:]
ops.qk_prep(q, H, hd, out_scale=ops.q_scale(hd))
vt = ops.prepare_v(v, H, hd)
x = torch.randn(L, H * hd, generator=g, device="cuda", dtype=torch.float32)
sc = torch.randn(H * hd, generator=g, device="cuda") * 0.1
for _ in range(iters):
    ops.attention(q, k, v, H, hd, v_prepared=vt, q_prescaled=True)
    ops.layernorm(x, scale=sc, shift=sc, eps=1e-6)
torch.cuda.synchronize()
print("done", iters)
