"""What a streaming kernel can reach on this chip at the LayerNorm's sizes: torch's own fp32 -> bf16 cast (read 4 B, write 2 B per element)
and a bf16 copy, against fw_layernorm_mod (same traffic as the cast)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fantasy_world_amd.hip_ops import HipOps
ops = HipOps("cuda:0")
L, D = 32760, 5120
x = torch.randn(L, D, device="cuda")
y = torch.empty(L, D, device="cuda", dtype=torch.bfloat16)
z = torch.empty_like(y)
sc, sh = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
tc = t(lambda: y.copy_(x)); tb = t(lambda: z.copy_(y)); tl = t(lambda: ops.layernorm(x, scale=sc, shift=sh, eps=1e-6, out=y))
tq = t(lambda: ops.qk_prep(y, 40, 128, out_scale=0.5))
from fantasy_world_amd import rope as _rope
tab = ops.to_f32(_rope.rope3d_table(128, 21, 30, 52))
nw = torch.randn(D, device="cuda")
tqr = t(lambda: ops.qk_prep(y, 40, 128, norm="rms_full", norm_w=nw, eps=1e-6, rope="interleaved", table=tab, out_scale=0.5))
print(f"fw_qk_prep (full-width RMSNorm + RoPE-3D, the DiT q / k): {tqr*1e3:.0f} us = {L*D*4/tqr/1e9:.2f} TB/s")
print(f"torch fp32->bf16 cast: {tc*1e3:.0f} us = {L*D*6/tc/1e9:.2f} TB/s | torch bf16 copy: {tb*1e3:.0f} us = {L*D*4/tb/1e9:.2f} TB/s | "
      f"fw_layernorm_mod: {tl*1e3:.0f} us = {L*D*6/tl/1e9:.2f} TB/s | fw_qk_prep (in place, scale only): {tq*1e3:.0f} us = {L*D*4/tq/1e9:.2f} TB/s")
