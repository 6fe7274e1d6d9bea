#!/usr/bin/env python
"""Workload + summary of tools/clock_probe.sh: a few launches of each hot kernel (and of torch.matmul as the vendor yardstick), then --
from the rocprofv3 database -- per kernel: average duration, GRBM_GUI_ACTIVE per dispatch / duration = effective clock (the counter is
reported per shader engine or summed: the ratio against a kernel of known behaviour is what matters), SQ_VALU_MFMA_BUSY / SQ_BUSY."""
import argparse, os, sys
ap = argparse.ArgumentParser()
ap.add_argument("--random", action="store_true"); ap.add_argument("--zeros", action="store_true")
ap.add_argument("--summarise"); ap.add_argument("--fill", default="")
args = ap.parse_args()
if args.summarise:
    import sqlite3, re
    from collections import defaultdict
    con = sqlite3.connect(args.summarise); cur = con.cursor()
    dur = defaultdict(lambda: [0, 0.0])
    for name, gx, wx, d in cur.execute("select name, grid_x, workgroup_x, duration from kernels"):
        dur[(name, gx // max(wx, 1))][0] += 1; dur[(name, gx // max(wx, 1))][1] += d
    cnt = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for name, gs, ws, cname, val in cur.execute("select kernel_name, grid_size, workgroup_size, counter_name, value from counters_collection"):
        c = cnt[(name, gs // max(ws, 1))][cname]; c[0] += 1; c[1] += val
    print(f"# fill = {args.fill}")
    print(f"{'kernel':58s} {'wgs':>6s} {'n':>3s} {'avg_us':>9s} {'GUI_ACTIVE/us = MHz x SEs':>26s} {'MFMA_BUSY/SQ_BUSY (of 32)':>26s}")
    for (name, wgs), (n, tot) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
        short = re.sub(r"\(anonymous namespace\)::|^void ", "", name).split("(")[0][:58]
        if short.startswith("at::") or n < 2 or tot / n < 50e3:
            continue
        c = cnt.get((name, wgs), {})
        # rows per dispatch differ per counter (dimensions): total value per dispatch = sum / dispatches
        per = {k: v[1] / n for k, v in c.items()}
        avg_us = tot / n / 1e3
        gui = per.get("GRBM_GUI_ACTIVE", 0.0) / avg_us if avg_us else 0.0
        mf = 32.0 * per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / per["SQ_BUSY_CYCLES"] / 32.0 if per.get("SQ_BUSY_CYCLES") else 0.0
        print(f"{short:58s} {wgs:6d} {n:3d} {avg_us:9.1f} {gui:26.1f} {mf:26.2f}")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fantasy_world_amd.hip_ops import HipOps, Linear
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
L = 32760
mk = (lambda *s: torch.zeros(*s, device="cuda", dtype=torch.bfloat16)) if args.zeros else (lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16))
x = mk(L, 5120); w = mk(15360, 5120) * (1.0 if args.zeros else 5120 ** -0.5)
lin = Linear(w.contiguous(), torch.zeros(15360, device="cuda")); out = torch.empty(L, 15360, dtype=torch.bfloat16, device="cuda")
q = mk(L, 2 * 128); k = mk(L, 2 * 128); v = mk(L, 2 * 128)
qs = (q.float() * ops.q_scale(128)).to(torch.bfloat16)
for rep in range(3):
    for kern in (4, 9):
        ops.set_option("gemm_kernel", kern)
        for _ in range(4): ops.linear(x, lin, out=out)
    for _ in range(4): torch.matmul(x, w.t(), out=out)
    for _ in range(4): ops.attention(qs.repeat(1, 4)[:, :1024].contiguous(), k.repeat(1, 4)[:, :1024].contiguous(), v.repeat(1, 4)[:, :1024].contiguous(), 8, 128, q_prescaled=True)
# round 5: the other attention kernels (hd 96 / hd 64 were never probed after the descriptor change) and fp8
L2 = 32865
q96 = mk(L, 12 * 96); k96 = mk(L2, 12 * 96); v96 = mk(L2, 12 * 96)
q96 = (q96.float() * ops.q_scale(96)).to(torch.bfloat16)
q64 = mk(L2, 16 * 64); k64 = mk(L2, 16 * 64); v64 = mk(L2, 16 * 64)
q64 = (q64.float() * ops.q_scale(64)).to(torch.bfloat16)
q8s = (q.float() * ops.q_scale_fp8(128)).to(torch.bfloat16).repeat(1, 4)[:, :1024].contiguous()
k8 = ops.cast_fp8(k.repeat(1, 4)[:, :1024].contiguous()); q8 = ops.cast_fp8(q8s)
vt8, lk8 = ops.prepare_v_fp8(v.repeat(1, 4)[:, :1024].contiguous(), 8, 128)
k96s = (k96.float() * ops.q_scale(96)).to(torch.bfloat16); q96r = mk(L, 12 * 96)
for rep in range(3):
    for _ in range(4): ops.attention(q96, k96, v96, 12, 96, q_prescaled=True)
    for _ in range(4): ops.attention(k96s, q96r, q96r, 12, 96, q_prescaled=True)
    for _ in range(4): ops.attention(q64, k64, v64, 16, 64, q_prescaled=True)
    for _ in range(4): ops.attention_fp8(q8, k8, vt8, 8, 128, lk8)
# round 5: the fp8 GEMM (qkv shape) and the new fp8 attention kernel as well
lin8 = ops.pack_linear_fp8(w.contiguous(), torch.zeros(15360, device="cuda"))
xq = ops.quantize_fp8_rows(x)
for rep in range(3):
    for _ in range(4): ops.linear(xq, lin8, out=out)
torch.cuda.synchronize()
ops.set_option("gemm_kernel", 9)
