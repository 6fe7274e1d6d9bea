"""Fixed cost of an attention work-group (256 query rows of one head): time against Lk at fixed Lq, per ROUND of 256 work-groups."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fantasy_world_amd.hip_ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
for (H, hd, Lq, acc) in [(40, 128, 32768, False), (40, 128, 32768, True), (12, 96, 32768, False), (16, 64, 32768, False)]:
    q = torch.randn(Lq, H * hd, device="cuda", generator=g).bfloat16()
    pts = []
    for Lk in (256, 512, 1024, 2048, 4096, 8192):
        k = torch.randn(Lk, H * hd, device="cuda", generator=g).bfloat16()
        v = torch.randn(Lk, H * hd, device="cuda", generator=g).bfloat16()
        vt = ops.prepare_v(v, H, hd)
        out = torch.zeros(Lq, H * hd, device="cuda", dtype=torch.bfloat16)
        fn = lambda: ops.attention(q, k, v, H, hd, v_prepared=vt, q_prescaled=True, out=out, accumulate=acc)
        ts = []
        for rep in range(5):
            fn(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(8): fn()
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 8)
        ts.sort()
        rounds = (Lq // 256) * H / 256
        pts.append((Lk // 64, ts[2] * 1e3 / rounds))
    n = len(pts); sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts)
    sxx = sum(p[0] ** 2 for p in pts); sxy = sum(p[0] * p[1] for p in pts)
    b_ = (n * sxy - sx * sy) / (n * sxx - sx * sx); a_ = (sy - b_ * sx) / n
    print(f"H={H} hd={hd} accumulate={acc}: " + ", ".join(f"Lk={64*s}: {t:.1f}" for s, t in pts) + " us/round")
    print(f"    fit: {a_:.2f} us fixed per work-group + {b_:.3f} us per 64-key tile")
