"""TEST / MEASUREMENT INFRASTRUCTURE ONLY -- the CPU baseline leg of bench.py (`cpu_baseline` in its JSON line).

Times the reference's PyTorch-CPU path for one denoise step on the host cores of the machine bench.py runs on, the way
SURVEY.md 8(d) / BASELINE.md section 3 prescribe: the full fp32 model is 64 GB and takes hours per step, so one block of each kind is
timed at the workload's token counts and the step is assembled as

    t_step = 2 (CFG) x [ 16 t_PCB + 24 (t_frame + t_DiT + t_global + t_bicross) ]        (16 PCB blocks, 24 IRG iterations)

  kind = "reference": the reference's OWN modules (DiTBlock wan_video_dit.py:254-321, VGGT Block vggt/layers/block.py:22-116,
         CrossModalityBiAttentionBlock fusion/layer/block.py:146-221), default-initialised, fp32 -- wherever oracle/ref_locate.py
         finds the reference: /root/reference in the build container, the staged bundle oracle/_ref/reference_py.tgz on the GPU box;
  kind = "port": oracle/fw_oracle.py's restatement of the same blocks (pinned to the reference by tests/test_oracle_pin.py) --
         only when neither is present.

Thread count: swept over {8, 16, 32, 64, nproc} on a 2048-token DiT block (oversubscribing a big host is several times slower
than 8-16 threads), the best one is used and reported.  Token counts: the full workload when the projected time fits
`budget_s`; otherwise whole latent frames are dropped (same h x w grid) and each block is extrapolated by ITS algorithmic FLOPs
(linears scale with tokens, attention with tokens squared) -- the sample that was actually timed is stated in the result.
Not part of the product: nothing under fantasy_world_amd/ imports this file.
"""
import os
import time

import torch

from fantasy_world_amd import config as fwc, synth
from oracle import fw_oracle


def _dit_flops(cfg, L, Lc, adapter):
    D, Fd = cfg.dim, cfg.ffn_dim
    macs = 4 * L * D * D + 2 * L * L * D + 2 * L * D * D + 2 * Lc * D * D + 2 * L * Lc * D + 2 * L * D * Fd
    if adapter:
        macs += L * (2048 * 2048 + D * 1024 + 1024 * 2048 + 2048 * 409 + 409 * D)
    return 2.0 * macs


def _vggt_flops(cfg, S, P, frame):
    C, Cm = cfg.vggt_dim, cfg.vggt_mlp
    L2 = S * P
    return 2.0 * (L2 * (4 * C * C + 2 * C * Cm) + (2 * S * P * P * C if frame else 2 * L2 * L2 * C))


def _bicross_flops(cfg, L, L2):
    D, C, Bd = cfg.dim, cfg.vggt_dim, cfg.bicross_dim
    return 2.0 * (L * 3 * D * Bd + L2 * 3 * C * Bd + 4 * L * L2 * Bd)


class _PortBlocks:
    """oracle/fw_oracle.py block functions on synthetic weights (kind = "port")."""
    kind = "port"

    def __init__(self, cfg):
        self.cfg = one = fwc.FWConfig(num_layers=2, start_index=1, cross_attention_list=[0], has_image_input=cfg.has_image_input,
                                      camera_adapter=cfg.camera_adapter, control_adapter=cfg.control_adapter)
        spec = synth.weight_spec(one)
        self.p_dit = one.dit_prefix(0)
        self.p_frame = "vggt.aggregator.frame_blocks.0."
        self.p_glob = one.global_prefix(0)
        self.p_bi = "IRGBlock.0.bicross_attention."
        keep = (self.p_dit, self.p_frame, self.p_glob, self.p_bi)
        self.W = {n: synth.make_param(n, s, i) for n, (s, i) in spec.items() if n.startswith(keep)}
        self.adapter = one.has_adapter(0)

    def dit(self, x, ctx, t_mod, freqs, plucker):
        y, mods = fw_oracle.dit_block_partial(x, ctx, t_mod, freqs, self.W, self.p_dit, self.cfg, self.adapter, plucker)
        return fw_oracle.dit_block_remaining(y, mods, self.W, self.p_dit, self.cfg)

    def vggt(self, tok, pos, e0, frame):
        p = self.p_frame if frame else self.p_glob
        y, e = fw_oracle.vggt_block_partial(tok, pos, e0, self.W, p, self.cfg)
        return fw_oracle.vggt_block_remaining(y, e, self.W, p, self.cfg)

    def bicross(self, x, tok, fd, fa):
        return fw_oracle.bicross(x, tok, fd, fa, self.W, self.p_bi, self.cfg)


class _ReferenceBlocks:
    """The reference's own modules (kind = "reference"), default-initialised, fp32, eval mode."""
    kind = "reference"

    def __init__(self, cfg):
        from oracle import ref_harness
        ref_harness.install_stubs()
        flavour = "diffsynth_wan22" if cfg.control_adapter else "diffsynth_wan21"
        dit_mod = __import__(f"FantasyWorld.{flavour}.models.wan_video_dit", fromlist=["DiTBlock"])
        from FantasyWorld.vggt.layers.block import Block
        from FantasyWorld.vggt.layers.rope import RotaryPositionEmbedding2D
        from FantasyWorld.fusion.layer.block import CrossModalityBiAttentionBlock
        torch.manual_seed(0)
        self.cfg = cfg
        self.adapter = False          # the per-block camera adapter processor (1.1 % of the FLOPs) is not attached here
        self.dit_blk = dit_mod.DiTBlock(cfg.has_image_input, cfg.dim, cfg.num_heads, cfg.ffn_dim, cfg.eps).eval()
        mk = lambda: Block(dim=cfg.vggt_dim, num_heads=cfg.vggt_heads, mlp_ratio=4.0, qkv_bias=True, proj_bias=True,
                           ffn_bias=True, init_values=0.01, qk_norm=True,
                           rope=RotaryPositionEmbedding2D(frequency=cfg.vggt_rope_freq)).eval()
        self.frame_blk, self.glob_blk = mk(), mk()
        self.bi = CrossModalityBiAttentionBlock(cfg.dim, cfg.vggt_dim, cfg.bicross_dim, cfg.bicross_heads).eval()

    def dit(self, x, ctx, t_mod, freqs, plucker):
        return self.dit_blk(x, ctx, t_mod, freqs)

    def vggt(self, tok, pos, e0, frame):
        return (self.frame_blk if frame else self.glob_blk)(tok, pos=pos, e0=e0)

    def bicross(self, x, tok, fd, fa):
        return self.bi([x, tok], freqs=None, freqs_dit=fd, freqs_agg=fa)


def _inputs(cfg, f, h, w, Lc):
    g = torch.Generator().manual_seed(0)
    hw = h * w
    L, P = f * hw, cfg.n_special + hw
    hd, bhd = cfg.dim // cfg.num_heads, cfg.bicross_dim // cfg.bicross_heads
    fb = fw_oracle.precompute_freqs_cis_3d(bhd)
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    pos = torch.stack([ys.reshape(-1), xs.reshape(-1)], dim=-1) + 1
    pos = torch.cat([torch.zeros(cfg.n_special, 2, dtype=pos.dtype), pos], dim=0).unsqueeze(0).expand(f, -1, -1)
    return dict(
        x=torch.randn(1, L, cfg.dim, generator=g), ctx=torch.randn(1, Lc, cfg.dim, generator=g),
        t_mod=torch.randn(1, 6, cfg.dim, generator=g) * 0.1, e0=torch.randn(1, 6, cfg.vggt_dim, generator=g) * 0.1,
        plucker=torch.randn(1, L, cfg.plucker_dim, generator=g) if cfg.camera_adapter else None,
        tok=torch.randn(f, P, cfg.vggt_dim, generator=g), pos=pos,
        freqs=fw_oracle.expand_freqs(fw_oracle.precompute_freqs_cis_3d(hd), f, h, w),
        fd=fw_oracle.expand_freqs(fb, f, h, w), fa=fw_oracle.build_freqs_3d_with_extra_cis(fb, f, h, w, cfg.n_special),
        L=L, P=P, L2=f * P)


def _timed(fn):
    t0 = time.perf_counter()
    with torch.no_grad():
        fn()
    return time.perf_counter() - t0


def measure(cfg, F, h, w, budget_s=60.0, reference_root=None):
    """-> dict for bench.py's `cpu_baseline` (value in denoise-steps/s) plus the raw timings (profiles/rNN/cpu_baseline.json)."""
    from oracle import ref_locate
    root = reference_root or ref_locate.reference_root()
    blocks, ref_error = None, None
    if root and os.path.isdir(os.path.join(root, "FantasyWorld")):
        try:
            blocks = _ReferenceBlocks(cfg)
        except Exception as e:                     # a host that cannot import the reference must not cost the bench its line
            ref_error = f"{type(e).__name__}: {e}"[:300]
    if blocks is None:
        blocks = _PortBlocks(cfg)
    Lc = 512 + (cfg.clip_tokens if cfg.has_image_input else 0)
    ncpu = os.cpu_count() or 1
    hw = h * w

    # ---- thread sweep on a ~2048-token DiT block (one warm-up call first: thread pool, allocator, page-in) -------------
    fs = max(1, min(F, round(2048 / hw)))
    si = _inputs(cfg, fs, h, w, Lc)
    sweep_flops = _dit_flops(cfg, si["L"], Lc, blocks.adapter)
    run_dit = lambda i: blocks.dit(i["x"], i["ctx"], i["t_mod"], i["freqs"], i["plucker"])
    torch.set_num_threads(min(8, ncpu))
    _timed(lambda: run_dit(si))
    sweep = {}
    for n in sorted({min(t, ncpu) for t in (8, 16, 32, 64, ncpu)}):
        torch.set_num_threads(n)
        sweep[n] = sweep_flops / _timed(lambda: run_dit(si))
    threads = max(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    rate = sweep[threads]

    # ---- sample size: the full workload if its projected time fits the budget, else fewer latent frames ----------------
    def flops_at(f):
        L, P = f * hw, cfg.n_special + hw
        return dict(dit=_dit_flops(cfg, L, Lc, blocks.adapter), frame=_vggt_flops(cfg, f, P, True),
                    glob=_vggt_flops(cfg, f, P, False), bi=_bicross_flops(cfg, L, f * P))
    full = flops_at(F)
    f_s = F
    while f_s > 1 and sum(flops_at(f_s).values()) / rate > budget_s:
        f_s -= 1
    part = flops_at(f_s)
    i = _inputs(cfg, f_s, h, w, Lc)
    tg = i["tok"].reshape(1, i["L2"], -1)
    pg = i["pos"].reshape(1, i["L2"], 2)
    t = dict(dit=_timed(lambda: run_dit(i)),
             frame=_timed(lambda: blocks.vggt(i["tok"], i["pos"], i["e0"], True)),
             glob=_timed(lambda: blocks.vggt(tg, pg, i["e0"], False)),
             bi=_timed(lambda: blocks.bicross(i["x"], tg, i["fd"], i["fa"])))
    t_full = {k: t[k] * full[k] / part[k] for k in t}
    n_pcb, n_irg = cfg.start_index, cfg.n_irg
    t_step = 2.0 * (n_pcb * t_full["dit"] + n_irg * (t_full["frame"] + t_full["dit"] + t_full["glob"] + t_full["bi"]))
    eff = sum(part.values()) / sum(t.values())
    scaled = "" if f_s == F else (f"; timed at {f_s} of {F} latent frames ({i['L']} of {F * hw} DiT tokens) and extrapolated "
                                  "per block by its algorithmic FLOPs")
    what = ("the reference's own DiTBlock / VGGT Block (frame + global) / CrossModalityBiAttentionBlock" if blocks.kind == "reference"
            else "oracle port of DiTBlock (+ camera adapter) / VGGT Block (frame + global) / bicross block")
    return {
        "value": 1.0 / t_step, "unit": "denoise-steps/s", "cores": ncpu, "threads": threads, "kind": blocks.kind,
        "sample": (f"{what}, fp32 PyTorch-CPU, one block each at {i['L']} DiT / {i['L2']} VGGT tokens in "
                   f"{sum(t.values()):.1f} s ({eff / 1e12:.2f} TFLOP/s on {threads} threads, best of sweep "
                   f"{ {k: round(v / 1e12, 2) for k, v in sweep.items()} } TFLOP/s){scaled}; step = 2 x [{n_pcb} t_PCB + "
                   f"{n_irg} (t_frame + t_DiT + t_global + t_bicross)] = {t_step:.0f} s (embeddings, head"
                   + ("" if blocks.adapter else ", camera adapter") + " not counted)"),
        "raw": {"reference_import_error": ref_error, "seconds_timed": t, "seconds_full_size": t_full, "flops_timed": part, "flops_full_size": full,
                "thread_sweep_flops_per_s": {str(k): v for k, v in sweep.items()}, "frames_timed": f_s, "frames": F,
                "grid_hw": [h, w], "step_seconds": t_step},
    }
