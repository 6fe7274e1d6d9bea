"""Ragged-shape check of fw_attention_fp8 A/B arms against the default kernel: rows past Lq, a last tile with a handful of keys, one-tile
sequences -- the shapes on which an MFMA sunk under a partial EXEC showed (docs/kernels.md).  VARS=17,16,15,11,14 python tools/probes/fp8_attention_ragged_arms.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from fantasy_world_amd.hip_ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
hd = 128
VARS = [int(v) for v in os.environ.get("VARS", "17,16,15,11,14").split(",")]
for (B, H, Lq, Lk) in [(1, 1, 256, 200), (1, 1, 300, 192), (1, 1, 300, 200), (1, 1, 31, 7), (1, 1, 31, 64), (1, 1, 256, 7), (1, 2, 515, 1029)]:
    q, k, v = mk(B * Lq, H * hd), mk(B * Lk, H * hd), mk(B * Lk, H * hd)
    q8 = ops.cast_fp8((q.float() * ops.q_scale_fp8(hd)).to(torch.bfloat16)); k8 = ops.cast_fp8(k)
    vt8, lk = ops.prepare_v_fp8(v, H, hd, batch=B)
    outs = {}
    for var in [192] + VARS:
        ops.set_option("attn_var", var)
        outs[var] = ops.attention_fp8(q8, k8, vt8, H, hd, Lk, batch=B).float()
    ops.set_option("attn_var", 192)
    for var in VARS:
        d = (outs[var] - outs[192]).abs()
        rowerr = d.amax(1)
        badrows = (rowerr > 0.05 * outs[192].abs().max()).nonzero().flatten()
        print((B, H, Lq, Lk), var, "max|out|", outs[var].abs().max().item(), "ref", outs[192].abs().max().item(), "bad rows", len(badrows), badrows[:6].tolist(), badrows[-3:].tolist(),
              "ratio at worst", (outs[var].flatten()[d.argmax()] / outs[192].flatten()[d.argmax()]).item())
