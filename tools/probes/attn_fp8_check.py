"""fp8 attention (hd 128) against fp32 softmax attention and against the bf16 kernel; timing at the DiT self-attention shape."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fantasy_world_amd.hip_ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(7)
H, hd = 4, 128
def rel(a, b): return float((a - b).norm() / b.norm())
for (B, Lq, Lk) in [(1, 300, 200), (1, 256, 64), (2, 515, 1029), (1, 1024, 4096)]:
    q = torch.randn(B * Lq, H * hd, device="cuda", generator=g).bfloat16()
    k = torch.randn(B * Lk, H * hd, device="cuda", generator=g).bfloat16()
    v = torch.randn(B * Lk, H * hd, device="cuda", generator=g).bfloat16()
    qf = q.float().view(B, Lq, H, hd).permute(0, 2, 1, 3); kf = k.float().view(B, Lk, H, hd).permute(0, 2, 1, 3); vf = v.float().view(B, Lk, H, hd).permute(0, 2, 1, 3)
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) / math.sqrt(hd), -1) @ vf).permute(0, 2, 1, 3).reshape(B * Lq, H * hd)
    o16 = ops.attention(ops.qk_prep(q.clone(), H, hd, out_scale=ops.q_scale(hd)), k, v, H, hd, batch=B, q_prescaled=True).float()
    q8 = ops.cast_fp8(ops.qk_prep(q.clone(), H, hd, out_scale=ops.q_scale_fp8(hd)))
    k8 = ops.cast_fp8(k)
    vt8, _ = ops.prepare_v_fp8(v, H, hd, batch=B)
    o8 = ops.attention_fp8(q8, k8, vt8, H, hd, Lk, batch=B).float()
    torch.cuda.synchronize()
    print(f"B={B} Lq={Lq} Lk={Lk}: fp8 vs fp32 ref {rel(o8, ref):.3e}   bf16 kernel vs ref {rel(o16, ref):.3e}   finite {bool(torch.isfinite(o8).all())}", flush=True)
# timing at the DiT shape
H, L = 40, 32760
q = torch.randn(L, H * hd, device="cuda", generator=g).bfloat16(); k = torch.randn(L, H * hd, device="cuda", generator=g).bfloat16(); v = torch.randn(L, H * hd, device="cuda", generator=g).bfloat16()
q8 = ops.cast_fp8(ops.qk_prep(q.clone(), H, hd, out_scale=ops.q_scale_fp8(hd))); k8 = ops.cast_fp8(k); vt8, _ = ops.prepare_v_fp8(v, H, hd)
qs = ops.qk_prep(q.clone(), H, hd, out_scale=ops.q_scale(hd)); vt = ops.prepare_v(v, H, hd)
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
ops.set_option("attn_var", 8)
t8s = timeit(lambda: ops.attention_fp8(q8, k8, vt8, H, hd, L))
o8s = ops.attention_fp8(q8, k8, vt8, H, hd, L).float()
ops.set_option("attn_var", 192)
print(f"in-phase fp8 kernel: {t8s:.3f} ms")
t8 = timeit(lambda: ops.attention_fp8(q8, k8, vt8, H, hd, L))
t16 = timeit(lambda: ops.attention(qs, k, v, H, hd, v_prepared=vt, q_prescaled=True))
tc = timeit(lambda: (ops.cast_fp8(q), ops.cast_fp8(k), ops.prepare_v_fp8(v, H, hd)))
fl = 4.0 * L * L * H * hd
print(f"DiT self-attention: fp8 {t8:.3f} ms = {fl/t8/1e9:.0f} TF/s-equivalent; bf16 {t16:.3f} ms = {fl/t16/1e9:.0f} TF/s; casts + V transpose {tc:.3f} ms")
o8 = ops.attention_fp8(q8, k8, vt8, H, hd, L).float(); o16 = ops.attention(qs, k, v, H, hd, v_prepared=vt, q_prescaled=True).float()
print(f"full size: fp8 vs bf16 kernel rel-L2 {rel(o8, o16):.3e}; ping-pong vs in-phase fp8 kernel: equal {torch.equal(o8, o8s)} rel {rel(o8, o8s):.2e}")
