#!/usr/bin/env python
"""Where the cycles of the attention kernels go ("a counter table showing which stall reason owns the 23 %").

Workload (no arguments): a few launches of each attention kernel at the headline launch shapes, random data -- hd 128 self-attention
(8 of the 40 heads: same per-work-group work, 1024 work-groups), hd 96 bicross both directions, hd 64 VGGT global, the opt-in fp8
hd 128 kernel.  tools/attn_counters.sh runs it under rocprofv3 once per counter group (counters only, no tracing domains beside
--kernel-trace) and then calls `--summarise <db> ...` on every pass: per kernel and counter the per-dispatch total, and every
SQ wave-cycle counter as a fraction of SQ_WAVE_CYCLES (all of them count quad-cycles summed over waves, so the ratio is unit-free).
"""
import argparse, os, re, sys
ap = argparse.ArgumentParser()
ap.add_argument("--summarise", nargs="*")
args = ap.parse_args()
if args.summarise:
    import sqlite3
    from collections import defaultdict
    dur, cnt = defaultdict(lambda: [0, 0.0]), defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for path in args.summarise:
        cur = sqlite3.connect(path).cursor()
        seen = defaultdict(int)
        for name, gx, wx, d in cur.execute("select name, grid_x, workgroup_x, duration from kernels"):
            key = (name, gx // max(wx, 1))
            dur[key][0] += 1; dur[key][1] += d; seen[key] += 1
        per_pass = defaultdict(lambda: defaultdict(float))
        for name, gs, ws, cname, val in cur.execute("select kernel_name, grid_size, workgroup_size, counter_name, value from counters_collection"):
            per_pass[(name, gs // max(ws, 1))][cname] += val
        for key, cs in per_pass.items():
            for cname, tot in cs.items():
                c = cnt[key][cname]; c[0] += seen[key]; c[1] += tot
    order = ["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_WAVE_CYCLES",
             "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC",
             "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_INST_LEVEL_LDS", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT",
             "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_SALU", "SQ_INSTS_VMEM"]
    for key, (n, tot) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
        name, wgs = key
        short = re.sub(r"\(anonymous namespace\)::|^void ", "", name).split("(")[0][:60]
        if "attention" not in short or wgs < 256:
            continue
        per = {k: v[1] / max(v[0], 1) for k, v in cnt.get(key, {}).items()}
        avg_us = tot / n / 1e3
        print(f"\n== {short}  work-groups {wgs}  dispatches {n}  avg {avg_us:.1f} us")
        wc = per.get("SQ_WAVE_CYCLES")
        for c in order + sorted(set(per) - set(order)):
            if c not in per:
                continue
            extra = ""
            if c == "GRBM_GUI_ACTIVE":
                extra = f"   -> {per[c] / avg_us / 8 / 1e3:.3f} GHz effective shader clock (8 XCDs)"
            elif c == "SQ_VALU_MFMA_BUSY_CYCLES" and per.get("SQ_BUSY_CYCLES"):
                extra = f"   -> matrix pipe busy {per[c] / per['SQ_BUSY_CYCLES'] / 32:.3f} of SQ_BUSY_CYCLES x 32 SIMD-slots"
            elif c == "SQ_VALU_MFMA_COEXEC_CYCLES" and per.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                extra = f"   -> {per[c] / per['SQ_VALU_MFMA_BUSY_CYCLES']:.3f} of the MFMA-busy cycles also issued a plain VALU instruction"
            elif wc and c.startswith(("SQ_ACTIVE_INST", "SQ_WAIT")):
                extra = f"   -> {per[c] / wc:.3f} of the wave-cycles"
            elif c.startswith("SQ_INSTS_") and per.get("SQ_INSTS_MFMA"):
                extra = f"   -> {per[c] / per['SQ_INSTS_MFMA']:.2f} per MFMA"
            print(f"  {c:30s} {per[c]:18.0f}{extra}")
    sys.exit(0)

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fantasy_world_amd.hip_ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
L, L2 = 32760, 32865
q = mk(L, 8 * 128); k = mk(L, 8 * 128); v = mk(L, 8 * 128)
qs = (q.float() * ops.q_scale(128)).to(torch.bfloat16)
q96 = (mk(L, 12 * 96).float() * ops.q_scale(96)).to(torch.bfloat16); k96 = mk(L2, 12 * 96); v96 = mk(L2, 12 * 96)
k96s = (k96.float() * ops.q_scale(96)).to(torch.bfloat16); q96r = mk(L, 12 * 96)
q64 = (mk(L2, 16 * 64).float() * ops.q_scale(64)).to(torch.bfloat16); k64 = mk(L2, 16 * 64); v64 = mk(L2, 16 * 64)
q8 = ops.cast_fp8((q.float() * ops.q_scale_fp8(128)).to(torch.bfloat16)); k8 = ops.cast_fp8(k)
vt8, lk8 = ops.prepare_v_fp8(v, 8, 128)
FP8_ONLY = [int(x) for x in os.environ.get("FP8_VARS", "").split(",") if x]      # FP8_VARS=192,12: only the fp8 kernel, these FW_ATTN_VAR arms
for rep in range(2 if not FP8_ONLY else 0):
    for _ in range(3): ops.attention(qs, k, v, 8, 128, q_prescaled=True)
    for _ in range(3): ops.attention(q96, k96, v96, 12, 96, q_prescaled=True)
    for _ in range(3): ops.attention(k96s, q96r, q96r, 12, 96, q_prescaled=True)
    for _ in range(3): ops.attention(q64, k64, v64, 16, 64, q_prescaled=True)
    for _ in range(3): ops.attention_fp8(q8, k8, vt8, 8, 128, lk8)
for var in FP8_ONLY:
    ops.set_option("attn_var", var)
    for _ in range(6): ops.attention_fp8(q8, k8, vt8, 8, 128, lk8)
ops.set_option("attn_var", 192)
torch.cuda.synchronize()
