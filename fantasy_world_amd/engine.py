"""MI355X-native execution of FantasyWorldFusionModel.joint_forward (Wan2.1 flavour).

Host-side orchestration only: every arithmetic op is a call on `ops` (fantasy_world_amd.hip_ops.HipOps ->
libfw_mi355x.so).  The op decomposition follows the reference line by line:

  joint_forward            FantasyWorld/fusion/model_wan21.py:104-224
  DiTBlock (3 modes)       FantasyWorld/diffsynth_wan21/models/wan_video_dit.py:254-321
  Self/CrossAttention      wan_video_dit.py:159-243, camera adapter camera_control.py:92-148
  VGGT Block / Attention   FantasyWorld/vggt/layers/block.py:73-116, attention.py:50-72, rope.py:154-188
  IRGBlock / bicross       FantasyWorld/fusion/layer/block.py:43-94,179-221,532-625
  bridge / token assembly  FantasyWorld/vggt/models/vggt.py:118-131, aggregator.py:261-306
  Head / unpatchify        wan_video_dit.py:344-358,437-442

Numerics: residual streams (DiT x, VGGT tokens) are kept in fp32 (the reference keeps the DiT stream in bf16 and
the VGGT stream in fp32 after the first modulated block); matmul inputs are bf16, accumulation fp32; all
normalisation statistics, softmax and rotary math are fp32 (tables from fp64).
"""
import os
import types
from typing import Callable

import torch

from .config import FWConfig
from . import rope as _rope
from .parallel import Ready


def _pad_to(t, dim, size):
    if t.shape[dim] == size:
        return t
    shape = list(t.shape)
    shape[dim] = size - t.shape[dim]
    return torch.cat([t, torch.zeros(shape, dtype=t.dtype, device=t.device)], dim=dim)


def _ru64(n):
    return (n + 63) // 64 * 64


class _DitBlock:
    pass


class _VggtBlock:
    pass


class _Bicross:
    pass


class _Stage:
    """State carried between the begin / mid / end stages of an attention half."""
    pass


class _InvariantCache:
    """Step-invariant intermediates: values that depend only on tensors the caller passes unchanged
    through all 50 x 2 forwards of a generation -- the context embeddings (A2), every block's cross-attention K/V (A8), the
    camera adapter's Pluecker term (A9).  Keyed on the identity of the source tensors (address, shape, dtype, in-place version
    counter); the entry keeps the sources alive, so their storage cannot be recycled under the same address.  A few entries
    per tag (positive / negative prompt).  Disabled = every call recomputes, exactly like the reference."""

    def __init__(self, enabled, per_tag=4):
        self.enabled, self.per_tag, self.entries = enabled, per_tag, {}
        self.generation = 0          # bumped by clear(): whoever relies on "the cache is warm" (sampler.cfg_streams) re-checks

    @staticmethod
    def _ident(t):
        return (t.data_ptr(), tuple(t.shape), str(t.dtype), t._version, str(t.device))

    def get(self, tag, sources, fn):
        if not self.enabled:
            return fn()
        key = tuple(self._ident(t) for t in sources)
        slot = self.entries.setdefault(tag, [])
        for i, (k, val, _) in enumerate(slot):
            if k == key:
                slot.append(slot.pop(i))                    # most recently used last
                return val
        val = fn()
        slot.append((key, val, tuple(sources)))
        del slot[:-self.per_tag]
        return val

    def clear(self):
        self.entries.clear()
        self.generation += 1


class FusionEngine:
    """Holds the pre-packed weights of one fusion model on one device and runs joint_forward on it."""

    _mods = None          # (t_mod, DiT tables, e0, (VGGT tables, g1, g0)) of the forward in flight; None: stand-alone blocks (B2)
    _vggt_stack = ()

    def __init__(self, cfg: FWConfig, get: Callable[[str], torch.Tensor], ops, shard=None, heads_cfg=None,
                 cache_step_invariants=False, precision="bf16", fp8_attention=False):
        """`get(name)` returns the reference parameter `name` (any dtype/device); tensors are packed block by block
        so a 14B model never needs a second full-precision copy.  `shard` is an optional
        fantasy_world_amd.parallel.SequenceShard (one process per GPU, RCCL).  `heads_cfg` (config.HeadsConfig) enables the
        geometry heads: joint_forward(return_prediction=True) then returns the reference's prediction dict
        (vggt.py:134-154); their weights are packed on first use.  `cache_step_invariants` keeps the intermediates that
        only depend on the prompt / camera inputs across calls (same results, bit for bit; see _InvariantCache).
        `precision="fp8"` routes the DiT blocks' linears through the reference's fp8 linear; `fp8_attention=True` additionally
        runs the DiT self-attention (hd 128, 41 % of a step's FLOPs) on e4m3 q / k / v / probabilities (BASELINE config 5; parity
        UNPINNED -- the reference defines no fp8 attention).  Under a sequence shard the head exchange then carries q | k | v as
        e4m3 BYTES (half the all-to-all payload) and every rank runs fw_attention_fp8 on the heads it received: the same bytes
        reach the same kernel, so the sharded forward returns the unsharded one's bits."""
        if precision not in ("bf16", "fp8"):
            raise ValueError(f"precision must be 'bf16' or 'fp8', got {precision!r}")
        if fp8_attention not in (False, True, "bicross", "all"):
            raise ValueError("fp8_attention: False | True (DiT self-attention) | 'bicross' (+ the bicross attention) | 'all' (+ the VGGT "
                             "frame / global attention on the head_dim-64 kernel); the last two are round-6 experiments, unsharded engine only")
        if fp8_attention and cfg.head_dim != 128:
            raise ValueError("fp8_attention needs head_dim 128")
        if fp8_attention and not hasattr(ops, "attention_fp8"):
            raise ValueError("fp8_attention needs the HIP op set (fw_attention_fp8); it has no CPU statement")
        self.fp8_attention = bool(fp8_attention)
        # 'bicross': ALSO the two directions of the bicross attention (hd 96) on e4m3 operands, laid out head-by-head with zero padding to
        # 128 bytes so that the hd-128 kernel runs them unchanged (VERDICT r05 missing 2 / next 2d; unsharded engine only)
        self.fp8_bicross = fp8_attention in ("bicross", "all") and shard is None
        # 'all': ALSO the VGGT frame / global attention (hd 64) on e4m3 operands (fw_attention_fp8's head_dim-64 kernel): at BASELINE
        # config 5 the global attention is 2.1 s of a 15.7 s step in bf16
        self.fp8_vggt = fp8_attention == "all" and shard is None and cfg.vggt_dim // cfg.vggt_heads == 64
        self.cfg = cfg
        self.ops = ops
        self.shard = shard
        self.precision = precision
        self.heads_cfg = heads_cfg
        self._heads = None
        self._get = get
        # head groups per DiT self-attention exchange under a sequence shard (1 = one exchange); FW_SP_EXCHANGE_GROUPS overrides
        self.exchange_groups = int(os.environ.get("FW_SP_EXCHANGE_GROUPS", "2"))
        self.bicross_head_exchange = True    # sequence shard: head all-to-all for the bicross when its 12 heads divide (else row all-gathers)
        # Off by default: bench.py measures the reference's per-step work.  install() turns it on for real generations.
        self.invariants = _InvariantCache(cache_step_invariants)
        self._tables = {}
        self._plucker_zero_cache = None
        self._nb, self._ctx_sources, self._img_sources = 1, (), ()
        g, lin, lin_cat = self._packers(get)

        pd = "pipe.dit."
        self.kpatch = _ru64(cfg.in_dim * 4)
        self.patch = lin(pd + "patch_embedding.weight", pd + "patch_embedding.bias", k_pad=self.kpatch)
        self.text0 = lin(pd + "text_embedding.0.weight", pd + "text_embedding.0.bias")
        self.text2 = lin(pd + "text_embedding.2.weight", pd + "text_embedding.2.bias")
        self.time0 = ops.pack_linear_f32(g(pd + "time_embedding.0.weight"), g(pd + "time_embedding.0.bias"))
        self.time2 = ops.pack_linear_f32(g(pd + "time_embedding.2.weight"), g(pd + "time_embedding.2.bias"))
        self.timep = ops.pack_linear_f32(g(pd + "time_projection.1.weight"), g(pd + "time_projection.1.bias"))
        if cfg.has_image_input:
            self.img_ln0 = (ops.to_f32(g(pd + "img_emb.proj.0.weight")), ops.to_f32(g(pd + "img_emb.proj.0.bias")))
            self.img1 = lin(pd + "img_emb.proj.1.weight", pd + "img_emb.proj.1.bias")
            self.img3 = lin(pd + "img_emb.proj.3.weight", pd + "img_emb.proj.3.bias")
            self.img_ln4 = (ops.to_f32(g(pd + "img_emb.proj.4.weight")), ops.to_f32(g(pd + "img_emb.proj.4.bias")))
        if cfg.control_adapter:
            ca = pd + "control_adapter."
            self.ctl_conv = lin(ca + "conv.weight", ca + "conv.bias")                                   # [D, 24*64*4]
            self.ctl_res1 = lin(ca + "residual_blocks.0.conv1.weight", ca + "residual_blocks.0.conv1.bias")   # [D, 9*D]
            self.ctl_res2 = lin(ca + "residual_blocks.0.conv2.weight", ca + "residual_blocks.0.conv2.bias")
        self.head_mod = ops.to_f32(g(pd + "head.modulation").reshape(2, cfg.dim))
        self.head = lin(pd + "head.head.weight", pd + "head.head.bias")
        # round 6: the head's LayerNorm keeps its bf16 rounding remainder (ops.layernorm_split) and the 5120 -> 64 head GEMM runs on
        # both parts -- that ONE store was a third of the forward's bf16 floor (per-site ablation, docs/parity.md: noise_pred 3.0e-3
        # -> 2.5e-3 for two tiny launches).  split_head_norm = False: the round-1..5 form (A/B, tests)
        self.split_head_norm = hasattr(ops, "layernorm_split")
        self.head_lo = type(self.head)(self.head.w, None) if self.split_head_norm else None      # same weights, no bias: + W lo
        self.dit = [self._pack_dit(b, g, lin, lin_cat) for b in range(cfg.num_layers)]

        self.proj = lin("vggt.projection_head.weight", "vggt.projection_head.bias")
        cam = g("vggt.aggregator.camera_token")[0]        # [2,1,C]
        reg = g("vggt.aggregator.register_token")[0]      # [2,4,C]
        self.special = ops.to_f32(torch.cat([cam, reg], dim=1))   # [2, n_special, C]
        ctp = "vggt.aggregator.CamTokenProjector.mlp."             # only used when the caller passes camera_token
        self.camtok0 = lin(ctp + "0.weight", ctp + "0.bias", k_pad=64)          # 36 -> 128 (K padded to 64)
        self.camtok2 = lin(ctp + "2.weight", ctp + "2.bias")
        self.vtime0 = ops.pack_linear_f32(g("vggt.time_embedding.0.weight"), g("vggt.time_embedding.0.bias"))
        self.vtime2 = ops.pack_linear_f32(g("vggt.time_embedding.2.weight"), g("vggt.time_embedding.2.bias"))
        self.vtimep = ops.pack_linear_f32(g("vggt.time_projection.1.weight"), g("vggt.time_projection.1.bias"))
        self.frame = [self._pack_vggt(f"vggt.aggregator.frame_blocks.{j}.", g, lin) for j in range(cfg.n_irg)]
        self.glob = [self._pack_vggt(cfg.global_prefix(j), g, lin) for j in range(cfg.n_irg)]
        self.bicross = [self._pack_bicross(f"IRGBlock.{j}.bicross_attention.", g, lin, lin_cat)
                        for j in range(len(cfg.cross_attention_list))]
        # the learned modulation tables of ALL blocks of a kind, stacked: one fw_modulation_tables launch per forward adds the time
        # projection to every block's table (and forms the VGGT blocks' fc2 scale / offset vectors) instead of ~330 small tensor-op
        # launches.  Order of the VGGT stack: frame blocks, then global blocks.
        self._stack_modulation()

    def _stack_modulation(self):
        ops = self.ops
        self.dit_mod_all = torch.stack([b.mod for b in self.dit]).contiguous()
        for i, b in enumerate(self.dit):
            b.mod_slot = i
        vb = self.frame + self.glob
        self.vggt_mod_all = torch.stack([b.mod for b in vb]).contiguous() if vb else None
        self.vggt_ls2_all = torch.stack([b.ls2 for b in vb]).contiguous() if vb else None
        for i, b in enumerate(vb):
            b.mod_slot = i
        self._vggt_stack = vb
        self._mods = None

    # ------------------------------------------------------------------------------------------------ packing
    def _packers(self, get):
        """(g, lin, lin_cat): fetch a reference parameter as fp32 / pack one nn.Linear / pack several as one fused projection."""
        ops_ = self.ops
        g = lambda n: get(n).detach().to(torch.float32)

        def lin(wname, bname=None, k_pad=None, n_pad=None, fp8=False):
            w = g(wname)
            w = w.reshape(w.shape[0], -1)
            b = g(bname) if bname else None
            if k_pad:
                w = _pad_to(w, 1, k_pad)
            if n_pad:
                w = _pad_to(w, 0, n_pad)
                if b is not None:
                    b = _pad_to(b, 0, n_pad)
            return ops_.pack_linear(w, b, fp8=True) if fp8 else ops_.pack_linear(w, b)

        def lin_cat(names, fp8=False):
            w = torch.cat([g(n + ".weight") for n in names], dim=0)
            b = torch.cat([g(n + ".bias") for n in names], dim=0)
            return ops_.pack_linear(w, b, fp8=True) if fp8 else ops_.pack_linear(w, b)

        return g, lin, lin_cat

    def _pack_dit(self, b, g, lin, lin_cat, prefix=None, adapter=None):
        """prefix / adapter given: a stand-alone block (fantasy_world_amd.blocks, boundary B2) instead of block b of the model."""
        cfg, ops = self.cfg, self.ops
        p = cfg.dit_prefix(b) if prefix is None else prefix
        blk = _DitBlock()
        blk.index = b
        blk.mod = ops.to_f32(g(p + "modulation").reshape(6, cfg.dim))
        # precision "fp8": the DiT block's nn.Linear modules are the ones enable_vram_management swaps for AutoWrappedLinear
        # (diffsynth_wan21/vram_management/layers.py:145-166, module_map {nn.Linear: AutoWrappedLinear}) -- q/k/v/o of both
        # attentions and the FFN go through the fp8 linear; norms, modulation, the camera adapter and everything outside the
        # DiT blocks (embeddings, head, VGGT, bicross) keep the bf16 path
        f8 = self.precision == "fp8"
        blk.qkv = lin_cat([p + "self_attn.q", p + "self_attn.k", p + "self_attn.v"], fp8=f8)
        blk.o = lin(p + "self_attn.o.weight", p + "self_attn.o.bias", fp8=f8)
        blk.norm_q = ops.to_f32(g(p + "self_attn.norm_q.weight"))
        blk.norm_k = ops.to_f32(g(p + "self_attn.norm_k.weight"))
        blk.cq = lin(p + "cross_attn.q.weight", p + "cross_attn.q.bias", fp8=f8)
        blk.ckv = lin_cat([p + "cross_attn.k", p + "cross_attn.v"], fp8=f8)
        blk.co = lin(p + "cross_attn.o.weight", p + "cross_attn.o.bias", fp8=f8)
        blk.cnorm_q = ops.to_f32(g(p + "cross_attn.norm_q.weight"))
        blk.cnorm_k = ops.to_f32(g(p + "cross_attn.norm_k.weight"))
        if cfg.has_image_input:
            blk.ckv_img = lin_cat([p + "cross_attn.k_img", p + "cross_attn.v_img"], fp8=f8)
            blk.cnorm_k_img = ops.to_f32(g(p + "cross_attn.norm_k_img.weight"))
        blk.adapter = cfg.has_adapter(b) if adapter is None else bool(adapter)
        if blk.adapter:
            a = p + "cross_attn.processor."
            rp = _ru64(cfg.adapter_reduced)
            blk.a_g1 = lin(a + "k_proj.group1.weight", a + "k_proj.group1.bias")
            blk.a_g20 = lin(a + "k_proj.group2.0.weight", a + "k_proj.group2.0.bias")
            blk.a_g22 = lin(a + "k_proj.group2.2.weight", a + "k_proj.group2.2.bias")
            blk.a_v0 = lin(a + "v_proj.group2.0.weight", a + "v_proj.group2.0.bias", n_pad=rp)   # 409 -> 448 rows (zeros)
            blk.a_v2 = lin(a + "v_proj.group2.2.weight", a + "v_proj.group2.2.bias", k_pad=rp)   # 409 -> 448 cols (zeros)
        blk.norm3_w = ops.to_f32(g(p + "norm3.weight"))
        blk.norm3_b = ops.to_f32(g(p + "norm3.bias"))
        blk.ffn0 = lin(p + "ffn.0.weight", p + "ffn.0.bias", fp8=f8)
        blk.ffn2 = lin(p + "ffn.2.weight", p + "ffn.2.bias", fp8=f8)
        return blk

    def _pack_vggt(self, p, g, lin):
        cfg, ops = self.cfg, self.ops
        blk = _VggtBlock()
        blk.mod = ops.to_f32(g(p + "modulation").reshape(6, cfg.vggt_dim))
        blk.norm1 = (ops.to_f32(g(p + "norm1.weight")), ops.to_f32(g(p + "norm1.bias")))
        blk.qkv = lin(p + "attn.qkv.weight", p + "attn.qkv.bias")
        blk.q_norm = (ops.to_f32(g(p + "attn.q_norm.weight")), ops.to_f32(g(p + "attn.q_norm.bias")))
        blk.k_norm = (ops.to_f32(g(p + "attn.k_norm.weight")), ops.to_f32(g(p + "attn.k_norm.bias")))
        blk.proj = lin(p + "attn.proj.weight", p + "attn.proj.bias")
        blk.ls1 = ops.to_f32(g(p + "ls1.gamma"))
        blk.norm2 = (ops.to_f32(g(p + "norm2.weight")), ops.to_f32(g(p + "norm2.bias")))
        blk.fc1 = lin(p + "mlp.fc1.weight", p + "mlp.fc1.bias")
        blk.fc2 = lin(p + "mlp.fc2.weight", p + "mlp.fc2.bias")
        blk.ls2 = ops.to_f32(g(p + "ls2.gamma"))
        return blk

    def _pack_bicross(self, p, g, lin, lin_cat):
        ops = self.ops
        bc = _Bicross()
        c = p + "cross_attn."
        bc.qv1 = lin_cat([c + "m1_proj", c + "values_m1_proj"])      # x_dit -> [q | v1]
        bc.kv2 = lin_cat([c + "m2_proj", c + "values_m2_proj"])      # x_agg -> [k | v2]
        bc.out1 = lin(c + "out_m1_proj.weight", c + "out_m1_proj.bias")
        bc.out2 = lin(c + "out_m2_proj.weight", c + "out_m2_proj.bias")
        if getattr(self, "fp8_bicross", False):
            # the fp8 attention hands back O with every head padded to 128 columns (zeros): the out-projections take K = heads * 128
            # with zero weight columns at the padding
            Hb, hb = self.cfg.bicross_heads, self.cfg.bicross_dim // self.cfg.bicross_heads

            def padded(wname, bname):
                w = g(wname)
                wp = torch.zeros(w.shape[0], Hb, 128, dtype=w.dtype, device=w.device)
                wp[:, :, :hb] = w.view(w.shape[0], Hb, hb)
                return ops.pack_linear(wp.view(w.shape[0], Hb * 128), g(bname))
            bc.out1p = padded(c + "out_m1_proj.weight", c + "out_m1_proj.bias")
            bc.out2p = padded(c + "out_m2_proj.weight", c + "out_m2_proj.bias")
        bc.gamma1 = ops.to_f32(g(p + "gamma_m1"))
        bc.gamma2 = ops.to_f32(g(p + "gamma_m2"))
        return bc

    # ------------------------------------------------------------------------------------------------ tables
    def _get_tables(self, f, h, w):
        key = (f, h, w)
        if key not in self._tables:
            cfg, ops = self.cfg, self.ops
            bhd = cfg.bicross_dim // cfg.bicross_heads
            t = dict(
                dit=ops.to_f32(_rope.rope3d_table(cfg.head_dim, f, h, w)),
                bi_dit=ops.to_f32(_rope.rope3d_table(bhd, f, h, w)),
                bi_agg=ops.to_f32(_rope.rope3d_table_with_extra(bhd, f, h, w, cfg.n_special)),
                vggt=ops.to_f32(_rope.rope2d_table(cfg.vggt_dim // cfg.vggt_heads, h, w, cfg.n_special,
                                                   cfg.vggt_rope_freq)),
            )
            self._tables = {key: t}     # keep only the latest grid
        return self._tables[key]

    # ------------------------------------------------------------------------------------------------ blocks
    # Attention halves are split into begin / mid / end so that, when the forward is sequence-sharded, the exchanges (Pending
    # objects: RCCL on its side stream) overlap independent work of the OTHER branch: the DiT q|k|v head exchange flies
    # while the VGGT frame block runs, the VGGT exchange while DiT attention runs, the output exchanges while the other
    # branch's attention / projections run.  Unsharded, every Pending is Ready and the order of independent ops is immaterial.
    def _dit_attn_begin(self, blk, x, t_mod, tabs):
        """LN + modulate, fused q|k|v projection, RMSNorm + RoPE; starts the head exchange (or the K/V all-gather)."""
        cfg, ops, sh = self.cfg, self.ops, self.shard
        D, H, hd = cfg.dim, cfg.num_heads, cfg.head_dim
        st = _Stage()
        st.blk, st.x, st.qkv8 = blk, x, None
        st.mod = self._dit_mod(blk, t_mod)                       # [6, D]: shift/scale/gate msa, shift/scale/gate mlp
        xn = ops.layernorm(x, scale=st.mod[1], shift=st.mod[0], eps=cfg.eps)
        qkv = ops.linear(xn, blk.qkv)
        q, k = qkv[:, :D], qkv[:, D:2 * D]
        tab = tabs["dit"] if sh is None else tabs["dit_local"]
        if self.fp8_attention:
            # e4m3 q | k written by the q/k pass itself (fw_qk_prep_fp8: the bits of cast(qk_prep(.)) without the two cast passes,
            # VERDICT r05 weak 4); q carries softmax_scale * log2(e) * 2^3.  Sharded: v is cast raw as well, so that the exchange
            # moves ONE byte per element; unsharded, v goes bf16 -> transposed e4m3 in one pass inside _dit_attn_mid.
            qkv8 = ops.empty(qkv.shape[0], 3 * D, dtype=torch.uint8)
            ops.qk_prep(q, H, hd, norm="rms_full", norm_w=blk.norm_q, eps=cfg.eps, rope="interleaved", table=tab,
                        out_scale=ops.q_scale_fp8(hd), out8=qkv8[:, :D])
            ops.qk_prep(k, H, hd, norm="rms_full", norm_w=blk.norm_k, eps=cfg.eps, rope="interleaved", table=tab, out8=qkv8[:, D:2 * D])
            if sh is not None:
                ops.cast_fp8(qkv[:, 2 * D:], out=qkv8[:, 2 * D:])
                qkv = None                                        # the bf16 projection is dead: only bytes travel
            st.qkv8 = qkv8
            wire = qkv8 if sh is not None else None
        else:
            # softmax_scale * log2(e) is folded into q before its bf16 rounding: attention then works in the log2 domain
            ops.qk_prep(q, H, hd, norm="rms_full", norm_w=blk.norm_q, eps=cfg.eps, rope="interleaved", table=tab, out_scale=ops.q_scale(hd))
            ops.qk_prep(k, H, hd, norm="rms_full", norm_w=blk.norm_k, eps=cfg.eps, rope="interleaved", table=tab)
            wire = qkv
        st.qkv = qkv
        st.exchange = sh is not None and sh.heads_divisible(H)
        if st.exchange:      # head exchange: my rows / all heads -> all rows / my heads, in groups of local heads
            st.groups = self._head_groups(H // sh.world)
            st.pend = [sh.rows_to_heads_async(wire, 3, sh.dit_counts, (a * hd, b * hd)) for a, b in st.groups]
        elif sh is not None:
            st.pend = sh.all_gather_rows_async(wire[:, D:], sh.dit_counts)      # [L, 2D] (k | v) of every rank
        else:
            st.pend = Ready(None)
        return st

    def _dit_mod(self, blk, t_mod):
        """Block table + time projection: a row of the per-forward table (fw_modulation_tables, one launch for all blocks) when this
        forward built one for the stacked blocks, else this block alone (stand-alone blocks of boundary B2)."""
        m = self._mods
        if m is not None and m[0] is t_mod and getattr(blk, "mod_slot", None) is not None and self.dit[blk.mod_slot] is blk:
            return m[1][blk.mod_slot]
        return self.ops.modulation_tables(blk.mod.unsqueeze(0), t_mod)[0]

    def _vggt_slot(self, blk):
        i = getattr(blk, "mod_slot", None)
        return i if (i is not None and self._mods is not None and self._mods[3] is not None and self._vggt_stack[i] is blk) else None

    def _vggt_mod(self, blk, e0):
        """[6, C] table of a VGGT block: modulation + e0 (block.py:73-81; e[2] is never used)."""
        i = self._vggt_slot(blk)
        if i is not None and self._mods[2] is e0:
            return self._mods[3][0][i]
        return self.ops.modulation_tables(blk.mod.unsqueeze(0), e0)[0]

    def _vggt_gates(self, blk, e):
        """The fc2 epilogue's per-column scale / offset: ls2 * (1 + e4) * e5 and ls2 * e3 * e5.  Taken from the per-forward table when
        `e` IS this block's row of it; formed by the same kernel otherwise (caller-supplied modifiers of boundary B2)."""
        i = self._vggt_slot(blk)
        if i is not None and e.data_ptr() == self._mods[3][0][i].data_ptr():
            return self._mods[3][1][i], self._mods[3][2][i]
        z = getattr(self, "_zero_row", None)
        if z is None or z.shape[1] != e.shape[1]:
            z = self._zero_row = torch.zeros(1, e.shape[1], dtype=torch.float32, device=e.device)
        _, g1, g0 = self.ops.modulation_tables(e.unsqueeze(0).contiguous(), z, blk.ls2.unsqueeze(0))      # e + 0 = e
        return g1[0], g0[0]

    def _dit_attn_mid(self, st):
        """Attention over the full key sequence; starts the inverse exchange of the output."""
        cfg, ops, sh = self.cfg, self.ops, self.shard
        D, H, hd = cfg.dim, cfg.num_heads, cfg.head_dim
        if st.exchange:
            # group g+1 is still travelling while group g is being attended to, and the output of group g travels back behind
            # the attention of group g+1: only the first q|k|v group and the last output group are exposed
            back = []
            for (a, b), pend in zip(st.groups, st.pend):
                got = pend.wait()
                if self.fp8_attention:     # the bytes the unsharded engine would hand the kernel for these heads: same bits out
                    vt8, Lk = ops.prepare_v_fp8(got[:, 2], b - a, hd)
                    o = ops.attention_fp8(got[:, 0], got[:, 1], vt8, b - a, hd, Lk)
                else:
                    o = ops.attention(got[:, 0], got[:, 1], got[:, 2], b - a, hd, q_prescaled=True)
                back.append(sh.heads_to_rows_async(o, sh.dit_counts))          # [L/n, world * (b-a) * hd]
            st.pend = back
        elif self.fp8_attention:
            # e4m3 q / k / v, fp32 scores and softmax, e4m3 probabilities (fw_attention_fp8; PARITY UNPINNED: the reference has
            # no fp8 attention, include/fw_mi355x.h)
            got = st.pend.wait()                                  # None, or the e4m3 k | v rows of every rank (all-gather fallback)
            k8, v = (st.qkv8[:, D:2 * D], st.qkv[:, 2 * D:]) if got is None else (got[:, :D], got[:, D:])
            vt8, Lk = ops.prepare_v_fp8(v, H, hd, batch=self._nb)
            st.pend = Ready(ops.attention_fp8(st.qkv8[:, :D], k8, vt8, H, hd, Lk, batch=self._nb))
        else:
            got = st.pend.wait()
            k, v = (st.qkv[:, D:2 * D], st.qkv[:, 2 * D:]) if got is None else (got[:, :D], got[:, D:])
            st.pend = Ready(ops.attention(st.qkv[:, :D], k, v, H, hd, batch=self._nb, q_prescaled=True))
        st.qkv = st.qkv8 = None

    def _dit_attn_end(self, st, ctx_txt, ctx_img, plucker):
        """o-projection into the stream, then cross-attention (+ camera adapter): DiTBlock.forward up to `return_partial`
        (wan_video_dit.py:296-306).  x: fp32 residual stream [L, D], updated in place.  Returns the modulation table."""
        cfg, ops = self.cfg, self.ops
        D, H, hd = cfg.dim, cfg.num_heads, cfg.head_dim
        blk, x, mod = st.blk, st.x, st.mod
        if st.exchange:
            if len(st.groups) == 1:
                o = st.pend[0].wait()
            else:                                   # columns back into head order: (rank, local head) = global head
                sh = self.shard
                o = ops.empty(x.shape[0], D)
                ov = o.view(x.shape[0], sh.world, D // sh.world)
                for (a, b), pend in zip(st.groups, st.pend):
                    ov[:, :, a * hd:b * hd] = pend.wait().view(x.shape[0], sh.world, (b - a) * hd)
        else:
            o = st.pend.wait()
        ops.linear(o, blk.o, g1=mod[2], res=x, out_f32=True, out=x)
        # cross-attention: text + image keys share q; outputs are summed (wan_video_dit.py:185-201)
        xn3 = ops.layernorm(x, w=blk.norm3_w, b=blk.norm3_b, eps=cfg.eps)
        qc = ops.linear(xn3, blk.cq)
        ops.qk_prep(qc, H, hd, norm="rms_full", norm_w=blk.cnorm_q, eps=cfg.eps, out_scale=ops.q_scale(hd))
        def text_kv():
            kv = ops.linear(ctx_txt, blk.ckv)
            ops.qk_prep(kv[:, :D], H, hd, norm="rms_full", norm_w=blk.cnorm_k, eps=cfg.eps)
            return kv

        def image_kv():
            kvi = ops.linear(ctx_img, blk.ckv_img)
            ops.qk_prep(kvi[:, :D], H, hd, norm="rms_full", norm_w=blk.cnorm_k_img, eps=cfg.eps)
            return kvi

        nb = self._nb
        # cached on the identity of the tensors the CALLER passes (ctx_txt / ctx_img are derived objects, rebuilt per call when merged)
        kv = self.invariants.get(("ckv", id(blk), nb), self._ctx_sources, text_kv)
        oc = ops.attention(qc, kv[:, :D], kv[:, D:], H, hd, batch=nb, q_prescaled=True)
        if ctx_img is not None:
            kvi = self.invariants.get(("ckv_img", id(blk), nb), self._img_sources, image_kv)
            ops.attention(qc, kvi[:, :D], kvi[:, D:], H, hd, batch=nb, out=oc, accumulate=True, q_prescaled=True)
        if blk.adapter and plucker is not None:
            # camera_control.py:109-127 ('adaln'): scale == 0 identically, so x <- x + shift
            t1 = ops.linear(oc, blk.a_g20, act="relu")
            pterm = self.invariants.get(("pterm", id(blk)), (plucker,), lambda: ops.linear(plucker, blk.a_g1))
            # the Pluecker term is a READ-ONLY residual operand (out != res is legal in fw_gemm_bf16): no copy of the cached
            # [L, 2048] tensor per block and forward; merged samples share the camera, so each sample's rows take the same residual
            # (rows do not depend on the launch that computes them: bit-identical to one launch over the stacked rows)
            Lr = pterm.shape[0]
            comb = ops.empty(nb * Lr, blk.a_g22.N)
            for i in range(nb):
                ops.linear(t1[i * Lr:(i + 1) * Lr], blk.a_g22, res=pterm, out=comb[i * Lr:(i + 1) * Lr])
            t2 = ops.linear(comb, blk.a_v0, act="relu")
            ops.linear(t2, blk.a_v2, res=oc, out=oc)
        ops.linear(oc, blk.co, res=x, out_f32=True, out=x)
        return mod

    def _head_groups(self, local_heads):
        """Split a rank's heads of the DiT self-attention into groups whose exchanges overlap each other's attention.  Groups
        hold an even number of heads (2 heads x 128 query blocks = one full round of the 256 CUs at 480p), so a launch on
        a group wastes no partial round: 10 -> (4, 6), 20 -> (10, 10); odd or small counts stay whole."""
        n = self.exchange_groups
        if n <= 1 or local_heads < 4 or local_heads % 2:
            return [(0, local_heads)]
        cuts = [0]
        for g in range(1, n):
            c = (local_heads * g // n) // 2 * 2
            if c > cuts[-1]:
                cuts.append(c)
        cuts.append(local_heads)
        return [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]

    def _dit_attn(self, blk, x, ctx_txt, ctx_img, t_mod, tabs, plucker):
        """Self-attention + cross-attention (+ camera adapter), nothing interleaved (the PCB blocks)."""
        st = self._dit_attn_begin(blk, x, t_mod, tabs)
        self._dit_attn_mid(st)
        return self._dit_attn_end(st, ctx_txt, ctx_img, plucker)

    def _dit_ffn(self, blk, x, mod):
        """FFN half (`run_remaining`, wan_video_dit.py:288-294)."""
        cfg, ops = self.cfg, self.ops
        xn = ops.layernorm(x, scale=mod[4], shift=mod[3], eps=cfg.eps)
        hbuf = ops.linear(xn, blk.ffn0, act="gelu_tanh")
        ops.linear(hbuf, blk.ffn2, g1=mod[5], res=x, out_f32=True, out=x)

    def _vggt_attn_begin(self, blk, tok, e0, tabs, batch, frame_mode):
        """LN + modulate, fused qkv, per-head LN + 2-D RoPE; global mode starts the head exchange (or the K/V all-gather)."""
        cfg, ops, sh = self.cfg, self.ops, self.shard
        C, H = cfg.vggt_dim, cfg.vggt_heads
        hd = C // H
        st = _Stage()
        st.blk, st.x, st.batch = blk, tok, batch
        st.mod = self._vggt_mod(blk, e0)                         # [6, C]; e[2] is never used (block.py:73-81)
        e = st.mod
        xn = ops.layernorm(tok, w=blk.norm1[0], b=blk.norm1[1], scale=e[1], shift=e[0], eps=cfg.vggt_eps)
        qkv = ops.linear(xn, blk.qkv)
        q, k = qkv[:, :C], qkv[:, C:2 * C]
        if getattr(self, "fp8_vggt", False):
            # round-6 experiment: q / k leave the per-head LayerNorm + RoPE pass as e4m3 (q with the 2^3 of the fp8 kernel), v is cast
            # by the transpose pass; both attention modes (frame: batch = frames) on fw_attention_fp8's head_dim-64 kernel
            qk8 = ops.empty(qkv.shape[0], 2 * C, dtype=torch.uint8)
            ops.qk_prep(q, H, hd, norm="ln_head", norm_w=blk.q_norm[0], norm_b=blk.q_norm[1], eps=cfg.vggt_eps,
                        rope="half2d", table=tabs["vggt"], out_scale=ops.q_scale_fp8(hd), out8=qk8[:, :C])
            ops.qk_prep(k, H, hd, norm="ln_head", norm_w=blk.k_norm[0], norm_b=blk.k_norm[1], eps=cfg.vggt_eps,
                        rope="half2d", table=tabs["vggt"], out8=qk8[:, C:])
            vt8, Lk = ops.prepare_v_fp8(qkv[:, 2 * C:], H, hd, batch=batch)
            st.qkv = None
            st.exchange = False
            st.pend = Ready(ops.attention_fp8(qk8[:, :C], qk8[:, C:], vt8, H, hd, Lk, batch=batch))
            st.fp8_done = True
            return st
        ops.qk_prep(q, H, hd, norm="ln_head", norm_w=blk.q_norm[0], norm_b=blk.q_norm[1], eps=cfg.vggt_eps,
                    rope="half2d", table=tabs["vggt"], out_scale=ops.q_scale(hd))
        ops.qk_prep(k, H, hd, norm="ln_head", norm_w=blk.k_norm[0], norm_b=blk.k_norm[1], eps=cfg.vggt_eps,
                    rope="half2d", table=tabs["vggt"])
        st.qkv = qkv
        shard_it = (not frame_mode) and sh is not None
        st.exchange = shard_it and sh.heads_divisible(H)
        if st.exchange:
            st.pend = sh.rows_to_heads_async(qkv, 3, sh.agg_counts)            # global attention: all tokens, my heads
        elif shard_it:
            st.pend = sh.all_gather_rows_async(qkv[:, C:], sh.agg_counts)
        else:
            st.pend = Ready(None)
        return st

    def _vggt_attn_mid(self, st):
        cfg, ops, sh = self.cfg, self.ops, self.shard
        C, H = cfg.vggt_dim, cfg.vggt_heads
        hd = C // H
        if getattr(st, "fp8_done", False):
            return
        got = st.pend.wait()
        if st.exchange:
            o = ops.attention(got[:, 0], got[:, 1], got[:, 2], H // sh.world, hd, batch=1, q_prescaled=True)
            st.pend = sh.heads_to_rows_async(o, sh.agg_counts)
        else:
            k, v = (st.qkv[:, C:2 * C], st.qkv[:, 2 * C:]) if got is None else (got[:, :C], got[:, C:])
            st.pend = Ready(ops.attention(st.qkv[:, :C], k, v, H, hd, batch=st.batch, q_prescaled=True))
        st.qkv = None

    def _vggt_attn_end(self, st):
        """x += ls1 * proj(attention) (block.py:73-76,99-107); returns the modulation table for the MLP half."""
        ops = self.ops
        o = st.pend.wait()
        ops.linear(o, st.blk.proj, g1=st.blk.ls1, res=st.x, out_f32=True, out=st.x)
        return st.mod

    def _vggt_attn(self, blk, tok, e0, tabs, batch, frame_mode):
        """Attention half of vggt Block.forward, nothing interleaved.  tok: fp32 [rows, C] in place."""
        st = self._vggt_attn_begin(blk, tok, e0, tabs, batch, frame_mode)
        self._vggt_attn_mid(st)
        return self._vggt_attn_end(st)

    def _vggt_mlp(self, blk, tok, e):
        """MLP half: x += (ls2(mlp(norm2(x)) * (1 + e4) + e3)) * e5 -- modulation AFTER the MLP (block.py:78-81)."""
        cfg, ops = self.cfg, self.ops
        xn = ops.layernorm(tok, w=blk.norm2[0], b=blk.norm2[1], eps=cfg.vggt_eps)
        hbuf = ops.linear(xn, blk.fc1, act="gelu_erf")
        g1, g0 = self._vggt_gates(blk, e)           # ls2 * (1 + e4) * e5 and ls2 * e3 * e5 (fw_modulation_tables, all blocks at once)
        ops.linear(hbuf, blk.fc2, g1=g1, g0=g0, res=tok, out_f32=True, out=tok)

    def _bicross(self, bc, x, tok, tabs):
        """CrossModalityBiAttentionBlock (fusion/layer/block.py:179-221) with BiMultiHeadAttention.forward_sdpa
        (block.py:532-625): both directions, residuals scaled by gamma_m1 / gamma_m2."""
        cfg, ops, sh = self.cfg, self.ops, self.shard
        Bd, Hb = cfg.bicross_dim, cfg.bicross_heads
        hd = Bd // Hb
        a = ops.layernorm(x, eps=1e-6)
        b = ops.layernorm(tok, eps=1e-6)
        qv1 = ops.linear(a, bc.qv1)          # [L, 2*Bd]  q | v1
        kv2 = ops.linear(b, bc.kv2)          # [L2, 2*Bd] k | v2
        # the scale goes on q only: direction 1 uses q as queries, direction 2 uses it as keys -- one factor either way
        nb = self._nb
        if getattr(self, "fp8_bicross", False):
            # both directions on fw_attention_fp8 (the hd-128 kernel on heads zero-padded from 96 to 128 bytes): q (carrying the softmax
            # scale x 2^3, undone by the kernel's operand scale whichever side of the product it sits on) and k written as e4m3 by the
            # rotary pass itself, v1 / v2 transposed + cast in one pass.  PARITY UNPINNED like every fp8 attention here: measured against
            # the bf16 bicross under the fp8 path's stated 2e-2 (docs/parity.md, round 6).
            q8 = ops.empty(qv1.shape[0], Hb * 128, dtype=torch.uint8)
            k8 = ops.empty(kv2.shape[0], Hb * 128, dtype=torch.uint8)
            ops.qk_prep(qv1[:, :Bd], Hb, hd, rope="interleaved", table=tabs["bi_dit"], out_scale=ops.q_scale_fp8(hd), out8=q8, head_stride8=128)
            ops.qk_prep(kv2[:, :Bd], Hb, hd, rope="interleaved", table=tabs["bi_agg"], out8=k8, head_stride8=128)
            vt1, L1 = ops.prepare_v_fp8(qv1[:, Bd:], Hb, hd, batch=nb, hd_out=128)
            vt2, L2k = ops.prepare_v_fp8(kv2[:, Bd:], Hb, hd, batch=nb, hd_out=128)
            o1 = ops.attention_fp8(q8, k8, vt2, Hb, 128, L2k, batch=nb)          # softmax(q k^T) v2
            o2 = ops.attention_fp8(k8, q8, vt1, Hb, 128, L1, batch=nb)           # softmax(k q^T) v1
            ops.linear(o1, bc.out1p, g1=bc.gamma1, res=x, out_f32=True, out=x)
            ops.linear(o2, bc.out2p, g1=bc.gamma2, res=tok, out_f32=True, out=tok)
            return
        ops.qk_prep(qv1[:, :Bd], Hb, hd, rope="interleaved", table=tabs["bi_dit"] if sh is None else tabs["bi_dit_local"],
                    out_scale=ops.q_scale(hd))
        ops.qk_prep(kv2[:, :Bd], Hb, hd, rope="interleaved", table=tabs["bi_agg"] if sh is None else tabs["bi_agg_local"])
        q_loc, k_loc = qv1[:, :Bd], kv2[:, :Bd]
        if sh is not None and self.bicross_head_exchange and sh.heads_divisible(Hb):
            # round 3: head exchange for the bicross too when the ranks divide its 12 heads (2, 3, 4, 6 ranks -- i.e. the 2 x 4 layout
            # of 8 GPUs): my rows / all heads -> all rows / my heads for q|v1 and k|v2 (two all-to-alls in flight together), both
            # directions for my heads over the full sequences, outputs back by the inverse exchange.  Each GPU receives (n-1)/n of
            # 2 x 38 MB + 2 x 19 MB per IRG block instead of (n-1)/n of 2 x 151 MB for the row all-gathers (4x fewer bytes at n = 4).
            Hl = Hb // sh.world
            p_kv2 = sh.rows_to_heads_async(kv2, 2, sh.agg_counts)
            p_qv1 = sh.rows_to_heads_async(qv1, 2, sh.dit_counts)
            gk, gq = p_kv2.wait(), p_qv1.wait()                  # [L2, 2, Hl*hd], [L, 2, Hl*hd]
            o1h = ops.attention(gq[:, 0], gk[:, 0], gk[:, 1], Hl, hd, batch=nb, q_prescaled=True)      # softmax(q k^T) v2, all DiT rows
            p_o1 = sh.heads_to_rows_async(o1h, sh.dit_counts)                                           # travels behind direction 2
            o2h = ops.attention(gk[:, 0], gq[:, 0], gq[:, 1], Hl, hd, batch=nb, q_prescaled=True)      # softmax(k q^T) v1, all VGGT rows
            p_o2 = sh.heads_to_rows_async(o2h, sh.agg_counts)
            o1, o2 = p_o1.wait(), p_o2.wait()
        else:
            if sh is not None:
                # both gathers in flight together; direction 1 only needs k|v2, so it runs while q|v1 is still arriving
                p_qv1 = sh.all_gather_rows_async(qv1, sh.dit_counts)
                p_kv2 = sh.all_gather_rows_async(kv2, sh.agg_counts)
            else:
                p_qv1, p_kv2 = Ready(qv1), Ready(kv2)
            kv2_all = p_kv2.wait()
            o1 = ops.attention(q_loc, kv2_all[:, :Bd], kv2_all[:, Bd:], Hb, hd, batch=nb, q_prescaled=True)      # softmax(q k^T) v2
            qv1_all = p_qv1.wait()
            o2 = ops.attention(k_loc, qv1_all[:, :Bd], qv1_all[:, Bd:], Hb, hd, batch=nb, q_prescaled=True)      # softmax(k q^T) v1
        ops.linear(o1, bc.out1, g1=bc.gamma1, res=x, out_f32=True, out=x)
        ops.linear(o2, bc.out2, g1=bc.gamma2, res=tok, out_f32=True, out=tok)

    # ------------------------------------------------------------------------------------------------ forward
    def _control_features(self, ctl, F, h, w):
        """Wan2.2 control adapter (wan_video_camera_controller.py:24-44,64-76): PixelUnshuffle(8) -> Conv2d(k2,s2) ->
        ResidualBlock(conv3x3 -> ReLU -> conv3x3, + skip), as three GEMMs over gathered patches.  fp32 [L, D], added to the
        patch embedding (wan_video_dit.py:390-396).  The input is constant over the whole generation (and identical for the
        positive and negative pass), the reference recomputes it in every call (31 TFLOP at 480p); with the step-invariant cache
        enabled it is computed once per control tensor (cache off = recomputed per call, like the reference)."""
        ops = self.ops

        def compute():
            c0 = ops.linear(ops.control_patchify(ctl), self.ctl_conv, out_f32=True)                     # conv k2 s2
            t1 = ops.linear(ops.im2col3x3(ops.cast_act(c0), F, h, w), self.ctl_res1, act="relu")        # conv1 + ReLU
            return ops.linear(ops.im2col3x3(t1, F, h, w), self.ctl_res2, res=c0, out_f32=True)          # conv2 + skip

        # the entry holds a reference to `ctl`, so its storage cannot be recycled under the same address by the next generation
        return self.invariants.get("control_features", (ctl,), compute)

    @torch.no_grad()
    def joint_forward(self, x, timestep, context, clip_feature=None, y=None, plucker_fea=None,
                      plucker_context_lens=None, uncond=False, return_prediction=False, camera_token=None,
                      control_camera_latents_input=None, collect=None):
        """Returns (noise_pred [1,16,F,H,W] in x.dtype, prediction).

        prediction is None unless return_prediction.  With `heads_cfg` it is the reference's prediction dict (pose_enc,
        depth, depth_conf, world_points, world_points_conf; vggt.py:134-154) computed by fantasy_world_amd.heads; without
        it, the aggregator's output_list as a dict layer -> fp32 [1, S, P, 2*C] for the layers the heads read.
        `collect`: optional dict that receives intermediate tensors (tests).
        """
        outs, pred = self._forward(x, timestep, [context], clip_feature, y, plucker_fea, plucker_context_lens, uncond,
                                   return_prediction, camera_token, control_camera_latents_input, collect)
        return outs[0], pred

    @torch.no_grad()
    def joint_forward_pair(self, x, timestep, context_pos, context_neg, clip_feature=None, y=None, plucker_fea=None,
                           plucker_context_lens=None, uncond=False, return_prediction=False, camera_token=None,
                           control_camera_latents_input=None):
        """The two CFG forwards of a sampling step in ONE pass ("CFG batch-2 merge"; the reference runs them
        one after the other, model_wan21.py:295-319, and diffsynth's own pipeline has the merged form,
        diffsynth_wan22/pipelines/wan_video_new.py:1576-1580).  The positive and the negative pass share latents, timestep, image
        / camera conditioning and differ ONLY in the text context, so every token-major op runs once on 2L rows (weights read
        once, twice the rows per launch) and the attentions run with batch 2.  Same arithmetic per row as two joint_forward calls:
        the results are bit-identical to them.  Returns (noise_pred_pos, noise_pred_neg, prediction of the positive pass)."""
        assert self.shard is None, "merged CFG is the single-GPU form; with several GPUs the two passes go to two rank groups"
        outs, pred = self._forward(x, timestep, [context_pos, context_neg], clip_feature, y, plucker_fea, plucker_context_lens,
                                   uncond, return_prediction, camera_token, control_camera_latents_input, None)
        return outs[0], outs[1], pred

    def _forward(self, x, timestep, contexts, clip_feature, y, plucker_fea, plucker_context_lens, uncond, return_prediction,
                 camera_token, control_camera_latents_input, collect):
        cfg, sh = self.cfg, self.shard
        st = self._prologue(x, timestep, contexts, clip_feature, y, plucker_fea, control_camera_latents_input)
        nb, P, tabs, xs, t_mod, e0 = st.nb, st.P, st.tabs, st.xs, st.t_mod, st.e0
        ctx_txt, ctx_img, plucker = st.ctx_txt, st.ctx_img, st.plucker
        # tests: collect["per_block"] = fn(kind, index, stream) is called with the fp32 streams after every block
        per_block = None if collect is None else collect.get("per_block")
        self._build_modulation(t_mod, e0)

        # ---- PCB: DiT blocks [0, start_index) ------------------------------------------------------------------
        for b in range(cfg.start_index):
            blk = self.dit[b]
            mod = self._dit_attn(blk, xs, ctx_txt, ctx_img, t_mod, tabs, plucker)
            self._dit_ffn(blk, xs, mod)
            if per_block is not None:
                per_block("x", b, xs)
        if collect is not None:
            collect["x_after_pcb"] = xs.clone()

        tok, S_loc = self._entry_tokens(st, camera_token)
        if collect is not None:
            collect["tokens_in"] = tok.clone()

        need = self._layers_for_heads(return_prediction)
        outputs = {}
        for i in range(cfg.n_irg):
            fb = self.frame[i]
            blk = self.dit[cfg.start_index + i]
            gb = self.glob[i]
            # The DiT block and the VGGT frame/global blocks of an IRG iteration are independent until the bicross
            # (fusion/layer/block.py:59-74), so their stages are interleaved to hide the exchanges of one branch behind the
            # compute of the other (same arithmetic as the reference order: frame block, DiT partial, VGGT global partial).
            sd = self._dit_attn_begin(blk, xs, t_mod, tabs)                     # DiT q|k|v exchange in flight ...
            e = self._vggt_attn(fb, tok, e0, tabs, batch=nb * S_loc, frame_mode=True)   # ... behind the VGGT frame block
            self._vggt_mlp(fb, tok, e)
            frame_out = tok[:S_loc * P].clone() if i in need else None                 # (merged: the positive sample's frames)
            sg = self._vggt_attn_begin(gb, tok, e0, tabs, batch=nb, frame_mode=False)   # VGGT exchange in flight ...
            self._dit_attn_mid(sd)                                              # ... behind DiT self-attention
            self._vggt_attn_mid(sg)                                             # DiT output exchange behind VGGT attention
            mod = self._dit_attn_end(sd, ctx_txt, ctx_img, plucker)             # VGGT output exchange behind DiT cross-attn
            e = self._vggt_attn_end(sg)
            if i in cfg.cross_attention_list and not uncond:
                self._bicross(self.bicross[cfg.cross_attention_list.index(i)], xs, tok, tabs)
            self._dit_ffn(blk, xs, mod)
            self._vggt_mlp(gb, tok, e)
            if per_block is not None:
                per_block("x", cfg.start_index + i, xs)
                per_block("tok", i, tok)
            if i in need:
                outputs[i] = torch.cat([frame_out.view(1, S_loc, P, -1), tok[:S_loc * P].view(1, S_loc, P, -1)], dim=-1)
        if collect is not None:
            collect["x_final"] = xs.clone()
            collect["tokens_final"] = tok.clone()
        return self._epilogue(st, x.dtype, outputs, return_prediction)

    def _build_modulation(self, t_mod, e0):
        """Two launches per forward: every DiT block's table + time projection; every VGGT block's table + e0 and fc2 vectors."""
        ops = self.ops
        self._mods = (t_mod, ops.modulation_tables(self.dit_mod_all, t_mod), e0,
                      None if self.vggt_mod_all is None else ops.modulation_tables(self.vggt_mod_all, e0, self.vggt_ls2_all))

    # The three parts of a forward that do not depend on how the blocks are partitioned (the tensor-parallel engine,
    # tensor_parallel.py, runs the same three around its own block loop).
    def _prologue(self, x, timestep, contexts, clip_feature, y, plucker_fea, control_camera_latents_input):
        """Time / text / image embeddings and the patchified latents: everything ahead of the first block.  Returns the forward's
        state: dims (F, h, w, hw, L, P, nb), tables, t / t_mod / e0, ctx_txt / ctx_img, plucker rows, and the fp32 stream xs."""
        cfg, ops, sh = self.cfg, self.ops, self.shard
        st = types.SimpleNamespace()
        nb = st.nb = self._nb = len(contexts)                  # samples stacked along the token rows (batch-major)
        assert x.shape[0] == 1, "the reference samples with batch 1 (model_wan21.py:254-258)"
        F, H2, W2 = x.shape[2:]
        h, w = H2 // 2, W2 // 2
        hw = h * w
        st.F, st.h, st.w, st.hw, st.L, st.P = F, h, w, hw, F * hw, cfg.n_special + hw
        tabs = self._get_tables(F, h, w)
        if sh is not None:
            tabs = sh.localize_tables(tabs, F, hw, cfg.n_special)
        st.tabs = tabs

        # ---- A1: time embeddings, fp32 (wan_video_dit.py:393-399, vggt.py:126-130) ----------------------------
        sin = ops.sinusoid(timestep, cfg.freq_dim)
        st.t = ops.linear_f32(ops.linear_f32(sin, self.time0, act="silu"), self.time2)        # [D]
        st.t_mod = ops.linear_f32(st.t, self.timep, silu_in=True).view(6, cfg.dim)
        ev = ops.linear_f32(ops.linear_f32(sin, self.vtime0, act="silu"), self.vtime2)
        st.e0 = ops.linear_f32(ev, self.vtimep, silu_in=True).view(6, cfg.vggt_dim)

        # ---- A2: context embeddings (wan_video_dit.py:388-392, 324-341) ----------------------------------------
        embs = [self.invariants.get("ctx_txt", (c,), lambda c=c: ops.linear(
            ops.linear(ops.to_act(c[0]), self.text0, act="gelu_tanh"), self.text2)) for c in contexts]
        st.ctx_txt = embs[0] if nb == 1 else torch.cat(embs, dim=0)                            # [nb * 512, D]
        self._ctx_sources, self._img_sources = tuple(contexts), (clip_feature,)
        st.ctx_img = None
        if cfg.has_image_input:
            def image_ctx():
                ci = ops.layernorm(ops.to_act(clip_feature[0]), w=self.img_ln0[0], b=self.img_ln0[1], eps=1e-5)
                ci = ops.linear(ops.linear(ci, self.img1, act="gelu_erf"), self.img3)
                return ops.layernorm(ci, w=self.img_ln4[0], b=self.img_ln4[1], eps=1e-5)
            st.ctx_img = self.invariants.get("ctx_img", (clip_feature,), image_ctx)
            if nb > 1:
                st.ctx_img = st.ctx_img.repeat(nb, 1)

        # ---- A3: patchify (Conv3d k=s=(1,2,2) as GEMM) ---------------------------------------------------------
        # Wan2.1 concatenates y when the DiT has image input (model_wan21.py:125-126), Wan2.2 whenever y is given (model_wan22.py:252-253)
        use_y = y is not None and (cfg.has_image_input or cfg.control_adapter)
        patches = ops.patchify(x, y if use_y else None, self.kpatch)                           # [L, 192] (x | y channels)
        ycam = None
        if cfg.control_adapter and control_camera_latents_input is not None:
            ycam = self._control_features(control_camera_latents_input, F, h, w)               # fp32 [L, D]
        st.plucker = None
        if plucker_fea is not None and cfg.camera_adapter:
            if not self._plucker_all_zero(plucker_fea):                                        # camera_control.py:111
                st.plucker = self.invariants.get("plucker_rows", (plucker_fea,), lambda: (
                    ops.to_act(plucker_fea[0]) if sh is None else sh.take_dit_rows(ops.to_act(plucker_fea[0]))))
        if sh is not None:
            patches = sh.take_dit_rows(patches)
            ycam = None if ycam is None else sh.take_dit_rows(ycam)
        xs = ops.linear(patches, self.patch, res=ycam, out_f32=True)                          # fp32 residual stream
        if nb > 1:
            xs = xs.repeat(nb, 1)             # the merged samples start from the same embedded latents; they diverge at the first cross-attention
        st.xs = xs
        return st

    def _entry_tokens(self, st, camera_token):
        """The bridge DiT tokens -> VGGT tokens (model_wan21.py:170-175): returns (fp32 token stream [nb * S_loc * P, C], S_loc)."""
        cfg, ops, sh = self.cfg, self.ops, self.shard
        nb, F, hw, P = st.nb, st.F, st.hw, st.P
        ptok = ops.linear(ops.cast_act(st.xs), self.proj)                                      # [L(local), C]
        if sh is not None:
            ptok = sh.dit_rows_to_frames(ptok, hw)                                             # this rank's frames
            S_loc = sh.my_frames
        else:
            S_loc = F
        if nb == 1:
            tok = ops.assemble_tokens(ptok, self._special_for(sh), S_loc, hw)                  # fp32 [S_loc*P, C]
        else:                                 # frame 0 of EVERY sample takes the first-frame camera / register tokens
            Ln = ptok.shape[0] // nb
            tok = torch.cat([ops.assemble_tokens(ptok[i * Ln:(i + 1) * Ln], self.special, S_loc, hw) for i in range(nb)], dim=0)
        if camera_token is not None:
            # CamTokenProjector (vggt/layers/block.py:276-297, aggregator.py:265-266): the learned camera token of every frame is
            # replaced by an MLP of 4 consecutive pose encodings (the sequence is padded with 3 copies of its first pose)
            ct = ops.to_act(camera_token[0])                                                   # [V, 9]
            ct = torch.cat([ct, ct[:1].expand(3, -1)], dim=0).reshape(-1, 36)                   # [S, 36]
            assert ct.shape[0] == F, (camera_token.shape, F)
            ct = _pad_to(ct, 1, 64)
            cam = ops.linear(ops.linear(ct, self.camtok0, act="gelu_erf"), self.camtok2, out_f32=True)     # fp32 [S, C]
            if sh is not None:
                cam = cam[sh.first_frame:sh.first_frame + S_loc]
            tok.view(nb * S_loc, P, cfg.vggt_dim)[:, 0, :] = cam if nb == 1 else cam.repeat(nb, 1)
        return tok, S_loc

    def _layers_for_heads(self, return_prediction):
        """The aggregator layers the geometry heads read: dpt_head.py:44 (23, 17, 11, 7) and camera_head.py:89 (the last one)."""
        if not return_prediction:
            return set()
        return set(self.heads_cfg.layer_idx if self.heads_cfg is not None else (7, 11, 17, 23)) | {self.cfg.n_irg - 1}

    def _epilogue(self, st, out_dtype, outputs, return_prediction):
        """Head (wan_video_dit.py:344-358) + unpatchify, and the prediction from the collected aggregator layers."""
        cfg, ops, sh = self.cfg, self.ops, self.shard
        F, h, w, L = st.F, st.h, st.w, st.L
        hm = ops.modulation_tables(self.head_mod.unsqueeze(0), st.t)[0]            # head.modulation + t (wan_video_dit.py:352-353)
        if self.split_head_norm:
            xn, xn_lo = ops.layernorm_split(st.xs, scale=hm[1], shift=hm[0], eps=cfg.eps)
            hd_out = ops.linear(xn, self.head, out_f32=True)                                   # [L(local), 64] = W hi + b
            ops.linear(xn_lo, self.head_lo, res=hd_out, out_f32=True, out=hd_out)              # + W lo
        else:
            xn = ops.layernorm(st.xs, scale=hm[1], shift=hm[0], eps=cfg.eps)
            hd_out = ops.linear(xn, self.head, out_f32=True)                                   # [L(local), 64]
        if sh is not None:
            hd_out = sh.all_gather_rows(hd_out, sh.dit_counts)
        outs = [ops.unpatchify(hd_out[i * L:(i + 1) * L], F, h, w, out_dtype) for i in range(st.nb)]
        if return_prediction:
            if sh is not None:
                outputs = {k: sh.gather_frames(v) for k, v in outputs.items()}
            if self.heads_cfg is None:
                return outs, outputs
            # once per generation; with a sequence shard every rank holds all frames here and computes the same dict
            return outs, self.geometry_heads().predict(outputs, F, h, w, patch_start_idx=cfg.n_special)
        return outs, None

    def geometry_heads(self):
        if self._heads is None:
            from .heads import GeometryHeads
            self._heads = GeometryHeads(self.heads_cfg, self._get, self.ops)
        return self._heads

    # ------------------------------------------------------------------------------------------------ helpers
    def _special_for(self, sh):
        if sh is None or sh.first_frame == 0:
            return self.special
        # ranks that do not own frame 0 only ever use the "other frames" variant (aggregator.py:283-306)
        return torch.stack([self.special[1], self.special[1]], dim=0).contiguous()

    def _plucker_all_zero(self, plucker_fea):
        """camera_control.py:111 (`plucker_fea.abs().sum() == 0` -> adapter skipped).  One host sync per NEW plucker tensor (the
        reference syncs 25x per forward); the verdict is remembered TOGETHER WITH the tensor, so a different tensor that the
        caching allocator places at the same address later (next generation) can never inherit it."""
        c = self._plucker_zero_cache
        if c is None or c[0] is not plucker_fea or c[1] != plucker_fea._version:
            self._plucker_zero_cache = c = (plucker_fea, plucker_fea._version, bool((plucker_fea == 0).all().item()))
        return c[2]
