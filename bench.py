#!/usr/bin/env python
"""denoise-steps/sec of FantasyWorld's joint_forward hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload: BASELINE.json configs[1] -- Wan2.1-I2V-14B-480P + IRG fusion + VGGT geometry branch, 81 frames x 480 x 832
(latents [1,16,21,60,104], L = 32760 DiT tokens, L2 = 32865 VGGT tokens), random weights of the real architecture
(40 DiT blocks, 24+24 VGGT blocks, 24 bicross blocks, 25 camera adapters), synthetic inputs.  One step = 2
joint_forward calls (CFG positive + negative, return_prediction=False) + CFG combine + flow-match Euler update
(FantasyWorld/fusion/model_wan21.py:289-322).  N > 1 (fantasy_world_amd/parallel.py): the two CFG forwards go to two rank
groups, each group sequence-shards its forward (head all-to-all for attention), i.e. strong scaling of one sample.

Prints ONE JSON line on rank 0 (see README / task contract) with `roofline` for the dominant kernel (the hd-128
self-attention launch, 41% of the forward's FLOPs) and, at N = 1, `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from fantasy_world_amd import config as fwc, synth                      # noqa: E402
from fantasy_world_amd.engine import FusionEngine                       # noqa: E402
from fantasy_world_amd.hip_ops import HipOps                            # noqa: E402
from fantasy_world_amd.parallel import init_topology                    # noqa: E402
from fantasy_world_amd.sampler import FlowMatchScheduler, denoise_step  # noqa: E402

MFMA_BF16_PEAK = 2.5e15      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md chip table


def forward_flops(cfg, L, L2, S, P, Lc, Li):
    """Algorithmic FLOPs of one joint_forward: 1 MAC = 2 FLOP, attention = QK^T + PV only."""
    D, Fd, C, Cm, Bd = cfg.dim, cfg.ffn_dim, cfg.vggt_dim, cfg.vggt_mlp, cfg.bicross_dim
    n_ad = sum(1 for b in range(cfg.num_layers) if cfg.has_adapter(b))
    n_irg, n_bi = cfg.n_irg, len(cfg.cross_attention_list)
    macs = cfg.num_layers * (4 * L * D * D + 2 * L * L * D + 2 * L * D * D + 2 * (Lc + Li) * D * D + 2 * L * (Lc + Li) * D + 2 * L * D * Fd)
    macs += n_ad * L * (2048 * 2048 + D * 1024 + 1024 * 2048 + 2048 * 409 + 409 * D)
    macs += 2 * n_irg * L2 * (4 * C * C + 2 * C * Cm) + n_irg * 2 * S * P * P * C + n_irg * 2 * L2 * L2 * C
    macs += n_bi * (L * 3 * D * Bd + L2 * 3 * C * Bd + 4 * L * L2 * Bd)
    macs += L * 144 * D + Lc * (cfg.text_dim * D + D * D) + Li * (1280 * 1280 + 1280 * D) + L * D * C + L * D * 64
    return 2.0 * macs


def cpu_baseline(cfg, step_flops):
    """Oracle ("port") timed on this box's host cores on a bounded sample: one full-width DiT block (self-attention +
    cross-attention + camera adapter + FFN, fp32, PyTorch CPU kernels) at a reduced token count, converted to
    denoise-steps/s through its algorithmic FLOP count (the full fp32 model is 64 GB and ~6600 s/step)."""
    from oracle import fw_oracle
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    Ls, Lc, Li = 3072, 512, cfg.clip_tokens
    one = fwc.FWConfig(num_layers=1, start_index=1, cross_attention_list=[])
    W = {}
    spec = synth.weight_spec(one)
    p = one.dit_prefix(0)
    for name, (shape, init) in spec.items():
        if name.startswith(p):
            W[name] = synth.make_param(name, shape, init)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, Ls, cfg.dim, generator=g)
    ctx = torch.randn(1, Li + Lc, cfg.dim, generator=g)
    t_mod = torch.randn(1, 6, cfg.dim, generator=g)
    pl = torch.randn(1, Ls, cfg.plucker_dim, generator=g)
    hd = cfg.dim // cfg.num_heads
    freqs = fw_oracle.expand_freqs(fw_oracle.precompute_freqs_cis_3d(hd), 3, 32, 32)
    D, Fd = cfg.dim, cfg.ffn_dim
    macs = (4 * Ls * D * D + 2 * Ls * Ls * D + 2 * Ls * D * D + 2 * (Lc + Li) * D * D + 2 * Ls * (Lc + Li) * D + 2 * Ls * D * Fd
            + Ls * (2048 * 2048 + D * 1024 + 1024 * 2048 + 2048 * 409 + 409 * D))
    t0 = time.time()
    with torch.no_grad():
        y, mods = fw_oracle.dit_block_partial(x, ctx, t_mod, freqs, W, p, one, True, pl)
        y = fw_oracle.dit_block_remaining(y, mods, W, p, one)
    dt = time.time() - t0
    eff = 2.0 * macs / dt
    return {"value": eff / step_flops, "unit": "denoise-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle DiT block (self+cross+adapter+FFN, fp32) at {Ls} tokens: {dt:.1f} s, "
                      f"{eff / 1e12:.3f} TFLOP/s effective, extrapolated by algorithmic FLOPs to the 81x480x832 step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=81)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--layers", type=int, default=40, help="debug only: anything but 40 is not the BASELINE workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    topo = init_topology()
    shard, rank, world, local = topo.shard, topo.rank, topo.world, topo.local
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    local = int(os.environ.get("FW_BENCH_DEVICE", local))     # debugging aid: several ranks on one GPU (with FW_DIST_BACKEND=gloo)
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    ops = HipOps(dev)

    cfg = fwc.wan21_14b()
    if args.layers != 40:
        cfg = fwc.FWConfig(num_layers=args.layers, start_index=min(16, args.layers - 1),
                           cross_attention_list=list(range(args.layers - min(16, args.layers - 1))))
    F = (args.frames - 1) // 4 + 1
    H2, W2 = args.height // 8, args.width // 8
    hw = (H2 // 2) * (W2 // 2)
    L, P = F * hw, cfg.n_special + hw
    L2 = F * P

    spec = synth.weight_spec(cfg)
    t0 = time.time()
    eng = FusionEngine(cfg, lambda n: synth.make_param(n, spec[n][0], spec[n][1], device=dev), ops, shard=shard)
    torch.cuda.synchronize()
    t_build = time.time() - t0

    ins = synth.make_inputs(cfg, F, H2, W2, seed=1, device=dev, dtype=torch.bfloat16)
    cond = dict(clip_feature=ins["clip_feature"], y=ins["y"], plucker_fea=ins["plucker_fea"],
                plucker_context_lens=ins["plucker_context_lens"])
    sched = FlowMatchScheduler()
    sched.set_timesteps(50)
    latents = ins["x"]

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    step_id = 0
    for _ in range(args.warmup):
        latents, _ = denoise_step(eng, sched, step_id, latents, ins["context"], ins["context_neg"], cond, topo=topo)
        step_id += 1
    ops.start_kernel_timing("attn_hd128_self", lambda kw: kw["hd"] == 128 and kw["Lk"] >= L)
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        latents, _ = denoise_step(eng, sched, step_id, latents, ins["context"], ins["context_neg"], cond, topo=topo)
        step_id += 1
    barrier()
    dt = time.time() - t0
    attn_ms, attn_n = ops.stop_kernel_timing()
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(latents.float()).all(), "non-finite latents"

    f_fwd = forward_flops(cfg, L, L2, F, P, 512, cfg.clip_tokens if cfg.has_image_input else 0)
    step_flops = 2.0 * f_fwd
    value = args.steps / dt
    # dominant kernel: one hd-128 self-attention launch = 4 * Lq_local * L * D FLOP (QK^T + PV)
    # (sharded: L rows x 40/n heads per rank after the head exchange = the same FLOPs as L/n rows x 40 heads)
    # under a sequence shard a block's heads go through the kernel in groups (FusionEngine._head_groups): average per launch
    n_groups = len(eng._head_groups(cfg.num_heads // topo.sp_world)) if shard is not None and shard.heads_divisible(cfg.num_heads) else 1
    attn_flops = 4.0 * L * L * cfg.dim / topo.sp_world / n_groups
    achieved = attn_flops / (attn_ms * 1e-3) if attn_n else 0.0
    # HBM-side traffic of the dominant kernel: measured with rocprofv3 PMC counters in separate passes (FETCH_SIZE, WRITE_SIZE)
    # as MI355X_MICROARCH.md prescribes, recorded under profiles/ with provenance; bench.py only reports the stored measurement
    # (a PMC pass cannot run inside the timed process).  Valid for the unsharded launch shape only.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")) as f:
            traffic = float(json.load(f)["attention_hd128_self"]["traffic_bytes_per_launch"]) if topo.sp_world == 1 else None
    except (OSError, KeyError, ValueError):
        traffic = None
    out = {
        "metric": "denoise-steps/sec (81x480x832 latents, 14B WanDiT + IRG + VGGT branch)",
        "value": value, "unit": "denoise-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "Wan2.1-I2V-14B-480P + IRG fusion + VGGT branch, 81f x 480 x 832, random weights"
                               if args.layers == 40 and (args.frames, args.height, args.width) == (81, 480, 832)
                               else f"DEBUG layers={args.layers} {args.frames}f x {args.height} x {args.width}",
                   "dit_tokens": L, "vggt_tokens": L2, "cfg_forwards_per_step": 2,
                   "parallelism": topo.describe(),
                   "tflop_per_step": step_flops / 1e12, "engine_build_s": round(t_build, 1)},
        "mfma_frac_whole_step": step_flops * value / (world * MFMA_BF16_PEAK),
        "roofline": {"bound": "mfma", "kernel": "attention_sp_kernel<128, 1> (DiT self-attention, one launch per block"
                               + ("" if n_groups == 1 else f", in {n_groups} head groups under the sequence shard") + ")",
                     "achieved": achieved / 1e12, "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                     "frac": achieved / MFMA_BF16_PEAK, "launches_timed": attn_n, "avg_launch_ms": attn_ms,
                     "flops_per_launch": attn_flops, "traffic": traffic,
                     "traffic_unit": "HBM-side bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01/pmc_traffic.json)",
                     "algorithmic_bytes_per_launch": 4.0 * L * cfg.dim * 2 / topo.sp_world / n_groups},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, step_flops)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
