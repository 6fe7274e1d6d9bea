#!/usr/bin/env python
"""Phase timeline of the two-slot ping-pong GEMM (TIMING build: FW_GEMM_KERNEL=9, var bit 1): s_memtime of one wave of each group of
work-group 0 over slabs 16..19 -- LOAD start | DMA issued | waits done (-> barrier) | MFMA start | MFMA end | vmcnt done (-> barrier)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fantasy_world_amd.hip_ops import HipOps, Linear
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
M, N, K = 32760, 5120, 5120
x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
lin = Linear(torch.randn(N, K, device="cuda", generator=g).to(torch.bfloat16) * K ** -0.5, torch.zeros(N, device="cuda"))
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
ops.set_option("gemm_kernel", 9)
for var in (0, 2):
    ops.set_option("gemm_var", var)
    for _ in range(3):
        ops.linear(x, lin, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        ops.linear(x, lin, out=out)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"== FW_GEMM_KERNEL=9 var {var}: {ms:.3f} ms, {2.0*M*N*K/ms/1e9:.0f} TF/s; {ms*1e6/800:.0f} ns per slab on average")
buf = (ctypes.c_ulonglong * 64)()
assert ops.lib.fw_debug_gemm_pp_timestamps(ctypes.cast(buf, ctypes.c_void_p), 64) == 0
t0 = min(buf[0], buf[32])
for grp in range(2):
    print(f"  group {'AB'[grp]}: slab | LOAD start | +reads,DMA issued | +waits | barrier -> MFMA start | +MFMA work | +vmcnt wait | (next LOAD start)")
    for sl in range(4):
        v = [buf[(grp * 4 + sl) * 8 + i] - t0 for i in range(6)]
        nxt = (buf[(grp * 4 + sl + 1) * 8] - t0) if sl < 3 else None
        print(f"    {16+sl} | {v[0]:6d} | +{v[1]-v[0]:4d} | +{v[2]-v[1]:4d} | {v[3]:6d} (+{v[3]-v[2]:4d}) | +{v[4]-v[3]:4d} | +{v[5]-v[4]:4d} | " + (f"{nxt:6d} (+{nxt-v[5]:4d})" if nxt is not None else ""))
print(f"  3 slabs, group A LOAD(16) start -> LOAD(19) start: {buf[3*8] - buf[0]} ticks")
ops.set_option("gemm_kernel", 9); ops.set_option("gemm_var", 0)
