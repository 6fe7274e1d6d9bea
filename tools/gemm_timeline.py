#!/usr/bin/env python
"""Phase timeline of the bf16 ping-pong GEMM (TIMING build, FW_GEMM_KERNEL=4 var bit 1): s_memtime at the start (barrier passed)
and at the end of the work of each of the four phases of slabs 16..19, work-group 0, one wave of each group."""
import ctypes
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fantasy_world_amd.hip_ops import HipOps, Linear

ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
M, N, K = 32760, 5120, 5120
x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
lin = Linear(torch.randn(N, K, device="cuda", generator=g).to(torch.bfloat16) * K ** -0.5, torch.zeros(N, device="cuda"))
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for var in (2,):
    ops.set_option("gemm_kernel", 4)
    ops.set_option("gemm_var", var)
    for _ in range(3):
        ops.linear(x, lin, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        ops.linear(x, lin, out=out)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    buf = (ctypes.c_ulonglong * 128)()
    assert ops.lib.fw_debug_gemm_timestamps(ctypes.cast(buf, ctypes.c_void_p), 128) == 0
    t0 = min(buf[0], buf[64])
    print(f"== FW_GEMM_KERNEL=4 TIMING build: {ms:.3f} ms, {2.0*M*N*K/ms/1e9:.0f} TF/s; "
          f"kernel = 10 rounds x 80 slabs -> {ms*1e6/800:.0f} ns per slab on average")
    names = ["LOAD0", "MFMA0", "LOAD1", "MFMA1"]
    for grp in range(2):
        print(f"  group {'AB'[grp]} (ticks since first stamp; per phase: start | end-of-work | -> next start = barrier wait)")
        for sl in range(4):
            row = []
            for ph in range(4):
                s_, e_ = buf[grp * 64 + sl * 8 + ph * 2] - t0, buf[grp * 64 + sl * 8 + ph * 2 + 1] - t0
                row.append(f"{names[ph]} {s_:6d}+{e_ - s_:4d}")
            print("    slab", 16 + sl, " | ".join(row))
    span = buf[3 * 8 + 6] - buf[0]
    print(f"  4 slabs (group A LOAD0(16) start -> MFMA1(19) start): {span} ticks")
ops.set_option("gemm_kernel", 4)
ops.set_option("gemm_var", 0)
