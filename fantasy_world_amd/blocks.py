"""Drop-in boundary B2: the reference's BLOCK-level call surfaces, rebound per module instance.

  install_dit_block(block)     DiTBlock.forward(x, context, t_mod, freqs, *, return_partial, run_remaining, modifiers, **kwargs)
                               FantasyWorld/diffsynth_wan21/models/wan_video_dit.py:279-313 (same in diffsynth_wan22 :248-277)
  install_vggt_block(block)    Block.forward(x, pos, e0, return_partial, run_remaining, modifiers)   vggt/layers/block.py:82-116
  install_irg_block(irg)       IRGBlock.forward(x_dit, x_agg, *, context, t_mod, freqs, freqs_dit, freqs_agg, pos, e0, uncond,
                               **kwargs) -> (x_dit, x_agg, [x_agg.view(B,-1,P,D)])                    fusion/layer/block.py:97-143
  install_blocks(model)        all of the above on a FantasyWorldFusionModel: its OWN joint_forward (model_wan21.py:104-224) then
                               walks its own Python loops and every block runs on the HIP kernels.

This is the survey's "fallback granularity": coarser than the op hooks (B3), finer than joint_forward (B1).  Each replacement
runs the SAME stage functions the engine runs (FusionEngine._dit_attn* / _dit_ffn / _vggt_attn* / _vggt_mlp / _bicross), so
everything inside a block stays fused (q|k|v projection, gate / LayerScale / residual epilogues, log2-domain attention).  What
a block-granular surface costs, and B1 does not pay: the residual stream crosses the boundary in the CALLER's dtype between
blocks (the reference keeps the DiT stream in bf16; the engine keeps fp32 inside a forward), the caller's complex rotary tables
are converted per new table object (cached on identity), and nothing is cached across steps.

Arguments keep the reference's meaning: `freqs*` are the complex tables the reference builds (model_wan21.py:132-147), `pos`
the integer (y, x) grid of the aggregator (aggregator.py:276-280), `t_mod` / `e0` the [1, 6, width] modulation inputs,
`modifiers` what `return_partial=True` handed out.  Batch 1 (the reference samples with batch 1, model_wan21.py:254-258).
Weights are SNAPSHOT at install time, as with install().  Every installer returns an `undo()` callable.
"""
import types

import torch

from .config import FWConfig
from .engine import FusionEngine, _InvariantCache


class _BlockRunner(FusionEngine):
    """The engine's stage functions without a model around them (no embeddings, no tables, no shard)."""

    def __init__(self, cfg, ops, precision="bf16"):            # deliberately NOT FusionEngine.__init__: nothing model-level
        self.cfg, self.ops, self.shard = cfg, ops, None
        self.precision, self.fp8_attention = precision, False
        self.heads_cfg = self._heads = None
        self.exchange_groups = 1
        self.invariants = _InvariantCache(False)
        self._tables = {}
        self._plucker_zero_cache = None
        self._nb, self._ctx_sources, self._img_sources = 1, (), ()


class _TableCache:
    """Caller-supplied rotary tables -> the engine's fp32 (cos, sin) layout, converted once per table OBJECT (the reference builds
    its tables once per joint_forward and hands the same objects to every block).  Entries hold their source, so an address is
    never recycled under a live key."""

    def __init__(self, ops, keep=8):
        self.ops, self.keep, self.entries = ops, keep, []

    def _lookup(self, src, build):
        for i, (s, val) in enumerate(self.entries):
            if s is src:
                self.entries.append(self.entries.pop(i))
                return val
        val = build()
        self.entries.append((src, val))
        del self.entries[:-self.keep]
        return val

    def rope_complex(self, freqs, rows, half):
        """complex [rows, 1, half] (e^{i angle}, wan_video_dit.py:80-102) -> fp32 [rows, half, 2] = (cos, sin)."""
        def build():
            f = freqs.reshape(rows, half).resolve_conj()
            return self.ops.to_f32(torch.view_as_real(f).to(torch.float32))
        return self._lookup(freqs, build)

    def rope2d_from_pos(self, pos, hd, base):
        """int (y, x) positions [..., 2] -> fp32 [rows, hd/2, 2]: vggt/layers/rope.py:82-131,154-188 -- pairs i < hd/4 turn the
        y half by pos_y * inv_freq[i], the next hd/4 the x half by pos_x * inv_freq[i]; angles in fp32 like the reference."""
        def build():
            half = hd // 2
            inv_freq = 1.0 / (base ** (torch.arange(0, half, 2, dtype=torch.float32) / half))
            p = pos.reshape(-1, 2).to(torch.float32)                   # on pos's device: no host round-trip
            inv_freq = inv_freq.to(p.device)
            ang = torch.cat([p[:, 0:1] * inv_freq[None], p[:, 1:2] * inv_freq[None]], dim=-1)
            return self.ops.to_f32(torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1))
        return self._lookup(pos, build)


def _default_ops(ops, device):
    if ops is None:
        from .hip_ops import HipOps
        ops = HipOps(device or "cuda")
    return ops


def _stream(ops, t):
    """A private fp32 copy of a residual stream [rows, width] (the stage functions update it in place; the reference's blocks
    never modify their input)."""
    return ops.to_f32(t).clone()


def _rebind(module, fn):
    original = module.forward
    module.forward = types.MethodType(fn, module)

    def undo():
        module.forward = original
    return undo


# ---------------------------------------------------------------------------------------------------------------- DiT block
def _dit_block_config(block, params, prefix=""):
    adapter = (prefix + "cross_attn.processor.k_proj.group1.weight") in params
    kw = {}
    if adapter:
        a = prefix + "cross_attn.processor."
        kw = dict(plucker_dim=params[a + "k_proj.group1.weight"].shape[0],
                  adapter_hidden=params[a + "k_proj.group2.0.weight"].shape[0],
                  adapter_reduced=params[a + "v_proj.group2.0.weight"].shape[0])
    return FWConfig(dim=block.dim, num_heads=block.num_heads, ffn_dim=block.ffn_dim, eps=block.norm1.eps,
                    has_image_input=bool(block.cross_attn.has_image_input), num_layers=1, start_index=1,
                    cross_attention_list=[], camera_adapter=adapter, **kw), adapter


class _DitSide:
    """One reference DiTBlock on the engine's stage functions."""

    def __init__(self, block, runner, params, prefix, tables):
        self.r, self.tables = runner, tables
        g, lin, lin_cat = runner._packers(params.__getitem__)
        self.blk = runner._pack_dit(0, g, lin, lin_cat, prefix=prefix, adapter=runner.cfg.camera_adapter)
        self.mod_dtype = params[prefix + "modulation"].dtype

    def split_context(self, context):
        """[1, (257 +) Lc, D] embedded context -> (text rows, image rows | None): CrossAttentionProcessor, wan_video_dit.py:187-191."""
        ops, cfg = self.r.ops, self.r.cfg
        c = ops.to_act(context[0])
        if cfg.has_image_input:
            return c[cfg.clip_tokens:].contiguous(), c[:cfg.clip_tokens].contiguous()
        return c, None

    def plucker_rows(self, kwargs):
        pf = kwargs.get("plucker_fea")
        if pf is None or not self.blk.adapter or self.r._plucker_all_zero(pf):          # camera_control.py:111
            return None
        return self.r.ops.to_act(pf[0])

    def mods_from(self, modifiers):
        """(shift_mlp, scale_mlp, gate_mlp) as handed out by return_partial -> the [6, D] table the FFN stage indexes."""
        D = self.r.cfg.dim
        rows = [t.detach().reshape(D).to(torch.float32) for t in modifiers]
        return self.r.ops.to_f32(torch.stack([torch.zeros_like(rows[0])] * 3 + rows, dim=0))

    def mods_out(self, mod):
        D = self.r.cfg.dim
        return tuple(mod[i].view(1, 1, D).to(self.mod_dtype) for i in (3, 4, 5))

    def attn_half(self, xs, context, t_mod, freqs, kwargs):
        r, cfg = self.r, self.r.cfg
        L = xs.shape[0]
        tabs = {"dit": self.tables.rope_complex(freqs, L, cfg.head_dim // 2)}
        ctx_txt, ctx_img = self.split_context(context)
        r._ctx_sources = r._img_sources = (context,)
        return r._dit_attn(self.blk, xs, ctx_txt, ctx_img, r.ops.to_f32(t_mod.reshape(6, cfg.dim)), tabs, self.plucker_rows(kwargs))


def install_dit_block(block, ops=None, device=None, precision="bf16", tables=None):
    """Rebind `block.forward` of a reference DiTBlock (all three modes: full / return_partial / run_remaining).
    `tables`: a _TableCache shared between blocks (install_blocks), so a forward's rotary tables are converted once."""
    ops = _default_ops(ops, device)
    params = dict(block.named_parameters())
    cfg, _ = _dit_block_config(block, params)
    side = _DitSide(block, _BlockRunner(cfg, ops, precision), params, "", tables or _TableCache(ops))

    def forward(self, x, context=None, t_mod=None, freqs=None, *, return_partial=False, run_remaining=False, modifiers=None,
                **kwargs):
        assert x.dim() == 3 and x.shape[0] == 1, "block-level drop-in: batch 1 (model_wan21.py:254-258)"
        xs = _stream(ops, x[0])
        back = lambda: xs.to(x.dtype)[None]
        if run_remaining:                                             # wan_video_dit.py:288-294
            assert modifiers is not None, "modifiers must provide"
            side.r._dit_ffn(side.blk, xs, side.mods_from(modifiers))
            return back()
        mod = side.attn_half(xs, context, t_mod, freqs, kwargs)       # :296-306
        if return_partial:
            return back(), side.mods_out(mod)
        if modifiers is not None:
            mod = side.mods_from(modifiers)
        side.r._dit_ffn(side.blk, xs, mod)                            # :311-313
        return back()

    return _rebind(block, forward)


# --------------------------------------------------------------------------------------------------------------- VGGT block
def _vggt_block_config(block):
    at = block.attn
    return dict(vggt_dim=at.qkv.in_features, vggt_heads=at.num_heads, vggt_mlp=block.mlp.fc1.out_features,
                vggt_eps=block.norm1.eps, vggt_rope_freq=float(getattr(at.rope, "base_frequency", 100.0)))


class _VggtSide:
    def __init__(self, block, runner, params, prefix, tables):
        self.r, self.tables = runner, tables
        g, lin, _ = runner._packers(params.__getitem__)
        self.blk = runner._pack_vggt(prefix, g, lin)

    def e0_rows(self, e0):
        C = self.r.cfg.vggt_dim
        if e0 is None:
            raise ValueError("block-level drop-in: the fusion path always modulates its VGGT blocks (e0 given, "
                             "aggregator.py:215-260); the un-modulated Block.forward stays the reference's")
        assert e0.shape[0] == 1, "one modulation for all frames (batch 1)"
        return self.r.ops.to_f32(e0.reshape(6, C))

    def attn_half(self, tok, pos, e0, batch):
        cfg = self.r.cfg
        tabs = {"vggt": self.tables.rope2d_from_pos(pos, cfg.vggt_dim // cfg.vggt_heads, cfg.vggt_rope_freq)}
        return self.r._vggt_attn(self.blk, tok, self.e0_rows(e0), tabs, batch=batch, frame_mode=True)

    def mods_from(self, modifiers):
        C = self.r.cfg.vggt_dim
        # one modulation for every frame (e0 has batch 1 and is repeated per frame, block.py:95-98): row 0 stands for all
        return torch.stack([self.r.ops.to_f32(t.reshape(-1, C)[0]) for t in modifiers], dim=0)

    def mods_out(self, e, B):
        C = self.r.cfg.vggt_dim
        return tuple(e[i].view(1, 1, C).expand(B, 1, C) for i in range(6))    # (modulation + e0).chunk(6, dim=1), block.py:95-101


def install_vggt_block(block, ops=None, device=None, tables=None):
    """Rebind `block.forward` of a reference VGGT Block (frame mode [S, P, C] and global mode [1, S*P, C] are the same call)."""
    ops = _default_ops(ops, device)
    params = dict(block.named_parameters())
    cfg = FWConfig(num_layers=1, start_index=1, cross_attention_list=[], **_vggt_block_config(block))
    side = _VggtSide(block, _BlockRunner(cfg, ops), params, "", tables or _TableCache(ops))

    def forward(self, x, pos=None, e0=None, return_partial=False, run_remaining=False, modifiers=None):
        B, N, C = x.shape
        tok = _stream(ops, x.reshape(B * N, C))
        # the reference's stream becomes fp32 at the first modulated block (bf16 + fp32 promotion, block.py:73-81)
        out_dtype = torch.promote_types(x.dtype, torch.float32 if (e0 is not None or modifiers is not None) else x.dtype)
        back = lambda: tok.to(out_dtype).view(B, N, C)
        if run_remaining:                                             # block.py:89-94
            assert modifiers is not None, "run_remaining need modifiers"
            side.r._vggt_mlp(side.blk, tok, side.mods_from(modifiers))
            return back()
        e = side.attn_half(tok, pos, e0, B)
        if return_partial:
            return back(), side.mods_out(e, B)
        if modifiers is not None:
            e = side.mods_from(modifiers)
        side.r._vggt_mlp(side.blk, tok, e)
        return back()

    return _rebind(block, forward)


# ---------------------------------------------------------------------------------------------------------------- IRG block
def install_irg_block(irg, ops=None, device=None, precision="bf16", tables=None):
    """Rebind `irg.forward` of a reference IRGBlock: DiT partial || VGGT-global partial -> bidirectional cross-attention ->
    DiT FFN || VGGT MLP (fusion/layer/block.py:43-94), all on the engine's stage functions."""
    ops = _default_ops(ops, device)
    params = dict(irg.named_parameters())
    dcfg, _ = _dit_block_config(irg.x_dit, params, "x_dit.")
    ca = irg.bicross_attention.cross_attn
    cfg = FWConfig(**{**dcfg.__dict__, **_vggt_block_config(irg.x_agg), "bicross_dim": ca.embed_dim, "bicross_heads": ca.num_heads})
    runner = _BlockRunner(cfg, ops, precision)
    tables = tables or _TableCache(ops)
    dit = _DitSide(irg.x_dit, runner, params, "x_dit.", tables)
    agg = _VggtSide(irg.x_agg, runner, params, "x_agg.", tables)
    g, lin, lin_cat = runner._packers(params.__getitem__)
    bc = runner._pack_bicross("bicross_attention.", g, lin, lin_cat)

    def forward(self, x_dit, x_agg, *, context, t_mod, freqs, freqs_dit, freqs_agg, pos=None, e0=None, uncond=False, **kwargs):
        assert x_dit.shape[0] == 1, "block-level drop-in: batch 1"
        S, P, C = x_agg.shape                                          # '(b s) p d', b = 1
        L = x_dit.shape[1]
        xs = _stream(ops, x_dit[0])
        tok = _stream(ops, x_agg.reshape(S * P, C))
        bh = cfg.bicross_dim // cfg.bicross_heads // 2
        tabs = {"dit": tables.rope_complex(freqs, L, cfg.head_dim // 2),
                "bi_dit": tables.rope_complex(freqs_dit, L, bh), "bi_agg": tables.rope_complex(freqs_agg, S * P, bh),
                "vggt": tables.rope2d_from_pos(pos, cfg.vggt_dim // cfg.vggt_heads, cfg.vggt_rope_freq)}
        ctx_txt, ctx_img = dit.split_context(context)
        runner._ctx_sources = runner._img_sources = (context,)
        mod = runner._dit_attn(dit.blk, xs, ctx_txt, ctx_img, ops.to_f32(t_mod.reshape(6, cfg.dim)), tabs,
                               dit.plucker_rows(kwargs))                                              # block.py:59-62
        e = runner._vggt_attn(agg.blk, tok, agg.e0_rows(e0), tabs, batch=1, frame_mode=False)        # :63-69 (global: one sequence)
        if uncond is not True:                                                                        # :70-76
            runner._bicross(bc, xs, tok, tabs)
        runner._dit_ffn(dit.blk, xs, mod)                                                             # :77-81
        runner._vggt_mlp(agg.blk, tok, e)                                                             # :83-87
        x_dit_out = xs.to(x_dit.dtype)[None]
        x_agg_out = tok.to(torch.promote_types(x_agg.dtype, torch.float32)).view(1, S * P, C)
        return x_dit_out, x_agg_out, [x_agg_out.view(1, -1, P, C)]

    return _rebind(irg, forward)


def install_blocks(model, ops=None, device=None, precision="bf16"):
    """Boundary B2 on a whole FantasyWorldFusionModel: every DiT block still in `pipe.dit.blocks` (the PCB blocks), every VGGT
    frame block and every IRGBlock is rebound; `model.joint_forward` stays the REFERENCE's (model_wan21.py:104-224).
    Returns an `undo()` callable."""
    ops = _default_ops(ops, device)
    tables = _TableCache(ops)              # one forward's tables (freqs, freqs_bi_*, pos) are converted once for all blocks
    undos = []
    for blk in model.pipe.dit.blocks:
        if hasattr(blk, "self_attn"):                                  # blocks moved into an IRGBlock are nn.Identity here
            undos.append(install_dit_block(blk, ops=ops, precision=precision, tables=tables))
    n_used = len(model.pipe.dit.blocks) - model.start_index            # frame blocks the fusion loop walks (model_wan21.py:185)
    for blk in list(model.vggt.aggregator.frame_blocks)[:n_used]:
        if hasattr(blk, "attn"):
            undos.append(install_vggt_block(blk, ops=ops, tables=tables))
    for blk in model.vggt.aggregator.global_blocks:                    # global blocks NOT moved into an IRGBlock
        if hasattr(blk, "attn"):
            undos.append(install_vggt_block(blk, ops=ops, tables=tables))
    for irg in model.IRGBlock:
        undos.append(install_irg_block(irg, ops=ops, precision=precision, tables=tables))

    def undo():
        for u in undos:
            u()
    return undo
