"""fw_attention_fp8 at head_dim 64 (round 6): against the fp32 softmax on ragged / one-tile / batched shapes, and timed against the bf16 hd-64
kernel on the VGGT global attention launch of the headline grid (16 heads, L2 = 32 865) and of config 5's (L2 = 111 755).
    python tools/probes/fp8_attention_hd64_check.py"""
import math, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fantasy_world_amd.hip_ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
hd = 64


def ref(q, k, v, B, H):
    Lq, Lk = q.shape[0] // B, k.shape[0] // B
    qf, kf, vf = (t.float().view(B, -1, H, hd).permute(0, 2, 1, 3) for t in (q, k, v))
    o = torch.softmax(qf @ kf.transpose(-1, -2) / math.sqrt(hd), -1) @ vf
    return o.permute(0, 2, 1, 3).reshape(B * Lq, H * hd)


for (B, H, Lq, Lk) in [(1, 2, 256, 64), (1, 2, 256, 128), (1, 3, 300, 200), (1, 1, 31, 7), (2, 3, 515, 1029), (1, 4, 1024, 4096), (1, 16, 1565, 1565), (3, 4, 133, 133), (1, 2, 700, 1)]:
    q, k, v = mk(B * Lq, H * hd), mk(B * Lk, H * hd), mk(B * Lk, H * hd)
    q8 = ops.cast_fp8((q.float() * ops.q_scale_fp8(hd)).to(torch.bfloat16)); k8 = ops.cast_fp8(k)
    vt8, lk = ops.prepare_v_fp8(v, H, hd, batch=B)
    got = ops.attention_fp8(q8, k8, vt8, H, hd, Lk, batch=B).float()
    want = ref(q, k, v, B, H)
    bf = ops.attention((q.float() * ops.q_scale(hd)).to(torch.bfloat16), k, v, H, hd, batch=B, q_prescaled=True).float()
    e = ((got - want).norm() / want.norm()).item()
    print(f"B {B} H {H} Lq {Lq} Lk {Lk}: fp8 hd64 vs fp32 softmax rel-L2 {e:.3e}  (bf16 kernel {((bf - want).norm() / want.norm()).item():.3e})  finite {bool(torch.isfinite(got).all())}", flush=True)

for L in (32865, 111755):
    H = 16
    q, k, v = mk(L, H * hd), mk(L, H * hd), mk(L, H * hd)
    q8 = ops.cast_fp8((q.float() * ops.q_scale_fp8(hd)).to(torch.bfloat16)); k8 = ops.cast_fp8(k)
    vt8, lk = ops.prepare_v_fp8(v, H, hd)
    qs = (q.float() * ops.q_scale(hd)).to(torch.bfloat16)
    vt = ops.prepare_v(v, H, hd)
    o8 = torch.empty(L, H * hd, dtype=torch.bfloat16, device="cuda"); ob = torch.empty_like(o8)
    fns = {"fp8 hd64": lambda: ops.attention_fp8(q8, k8, vt8, H, hd, lk, out=o8), "bf16 hd64": lambda: ops.attention(qs, k, None, H, hd, out=ob, v_prepared=vt, q_prescaled=True)}
    times = {n: [] for n in fns}
    for r in range(5):
        for n, fn in fns.items():
            fn(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3): fn()
            b.record(); torch.cuda.synchronize()
            times[n].append(a.elapsed_time(b) / 3)
    fl = 4.0 * L * L * H * hd
    print(f"# VGGT global, 16 heads x 64, L2 = {L}: {fl / 1e12:.2f} TFLOP; rel-L2 fp8 vs bf16 kernel {((o8.float() - ob.float()).norm() / ob.float().norm()).item():.3e}")
    for n, ts in times.items():
        ms = statistics.median(ts)
        print(f"  {n:10s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF/s")
