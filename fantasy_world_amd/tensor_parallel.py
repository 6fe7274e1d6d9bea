"""north_star's partition of the single forward: attention-HEAD / FFN-COLUMN tensor parallelism with all-reduce (RCCL over xGMI).

    FW_PARALLEL=tp python bench.py --gpus 8          # 2 CFG groups x TP 4   (parallel.make_topology(mode="tp"))
    FW_PARALLEL=sp python bench.py --gpus 8          # 2 CFG groups x 4-way sequence shard with head all-to-all (the default)

Both partitions run the same kernels on the same model; which one wins on the xGMI mesh is a measurement (docs/multi_gpu.md
gives the byte table that made the sequence shard the default; `bench.py`'s `comm` block reports the bytes and the exposed
time of either).  This module is the TP side:

  piece                    partition                                             collective per block (n ranks, config 2)
  DiT self-attention       heads 40 -> 40/n: q|k|v column-parallel, o row-parallel   all-reduce [L,5120] + [2,L] fp32 statistics
                           (the q/k RMSNorm spans all heads, wan_video_dit.py:170-171: per-rank sums of squares are all-reduced,
                            fw_row_sumsq -> fw_qk_prep_tp)
  DiT cross-attention      same head split, context k|v column-parallel              all-reduce [L,5120] + [L + ctx] fp32
  camera adapter           5120->1024 row-parallel (all-reduce before the ReLU),     all-reduce [L,1024]
                           409->5120 column-parallel (camera_control.py:24-63)
  DiT FFN                  13824 -> 13824/n columns, then row-parallel                all-reduce [L,5120]
  VGGT block (48)          heads 16 -> 16/n (per-head LayerNorm: no statistics),      2 x all-reduce [L2,1024]
                           MLP 4096 -> 4096/n
  bicross (24)             12 heads -> 12/n when n divides 12 (n = 2, 4);             all-reduce [L,5120] + [L2,1024]
                           otherwise (n = 8) projections replicated, attention        (fallback: 2 row all-gathers)
                           split by QUERY ROWS, updates all-gathered
  embeddings, norms,       replicated (the fp32 residual streams live on every rank)  --
  bridge, head

The fp32 residual streams are replicated; a row-parallel GEMM writes its partial sums (no bias), they are all-reduced in
`reduce_dtype` (bf16 by default: the 335 MB per [L,5120] message SURVEY.md counts; fp32 doubles the bytes), and the bias / gate /
LayerScale / residual epilogue runs after the reduction (fw_residual_add).  Overlap: every all-reduce is issued asynchronously
(RCCL's side stream) and waited for on the compute stream only where its result is consumed --
  * row-parallel GEMMs run in `chunks` row blocks: the reduction of block c flies behind the GEMM of block c+1;
  * inside an IRG iteration the DiT branch and the VGGT branch are independent until the bicross (fusion/layer/block.py:59-74):
    they are written as generators that yield after issuing a collective, and a round-robin driver advances the other branch
    meanwhile.
"""
import torch
import torch.distributed as dist

from .engine import FusionEngine, _DitBlock, _VggtBlock, _Bicross, _pad_to, _ru64
from .parallel import Pending, Ready, split_counts


class TensorShard:
    """This rank's place in one tensor-parallel group."""

    def __init__(self, rank, world, group=None, reduce_dtype=torch.bfloat16, chunks=2, chunk_rows=2048):
        self.rank, self.world, self.group = rank, world, group
        # row-parallel GEMMs run in up to `chunks` row blocks of at least `chunk_rows` rows (reduction of one behind the GEMM of the next)
        self.reduce_dtype, self.chunks, self.chunk_rows = reduce_dtype, chunks, chunk_rows

    def heads(self, H):
        if H % self.world:
            raise ValueError(f"{H} heads do not divide over {self.world} tensor-parallel ranks")
        n = H // self.world
        return self.rank * n, (self.rank + 1) * n

    def divides(self, H):
        return H % self.world == 0

    def units(self, N, unit=64):
        """[a, b) of N columns, split in whole `unit`s (GEMM K-slabs are 64 wide)."""
        assert N % unit == 0 and N // unit >= self.world, (N, unit, self.world)
        c = split_counts(N // unit, self.world)
        a = sum(c[: self.rank]) * unit
        return a, a + c[self.rank] * unit

    def rows(self, L):
        c = split_counts(L, self.world)
        a = sum(c[: self.rank])
        return a, a + c[self.rank], c

    def all_reduce_async(self, t, kind="all_reduce", op=None) -> Pending:
        """Sum (or `op`, e.g. MAX for the fp8 row maxima) over the group, in place on t (contiguous)."""
        assert t.is_contiguous()
        if self.world == 1:
            return Ready(t)
        work = dist.all_reduce(t, op=op or dist.ReduceOp.SUM, group=self.group, async_op=True)
        # bytes this rank sends in a bandwidth-optimal all-reduce (reduce-scatter + all-gather): 2 (n-1)/n of the message
        sent = 2 * t.numel() * t.element_size() * (self.world - 1) // self.world
        return Pending(work, lambda: t, kind, sent, t)

    def all_gather_rows_async(self, t, counts) -> Pending:
        t = t.contiguous()
        mx = max(counts)
        if self.world == 1:
            return Ready(t)
        pad = t if t.shape[0] == mx else torch.cat([t, t.new_zeros(mx - t.shape[0], t.shape[1])], dim=0)
        buf = torch.empty(self.world, mx, t.shape[1], dtype=t.dtype, device=t.device)
        work = dist.all_gather_into_tensor(buf.view(self.world * mx, t.shape[1]), pad, group=self.group, async_op=True)
        fin = (lambda: buf.view(-1, t.shape[1])) if min(counts) == mx else (
            lambda: torch.cat([buf[r, : counts[r]] for r in range(self.world)], dim=0))
        return Pending(work, fin, "all_gather_rows", pad.numel() * pad.element_size() * (self.world - 1), pad)


def _interleave(*gens):
    """Advance generators round-robin: each yields right after ISSUING a collective, so the other branches' compute is enqueued
    before the compute stream is made to wait for it."""
    active = list(gens)
    while active:
        for g in list(active):
            try:
                next(g)
            except StopIteration:
                active.remove(g)


def _drain(gen):
    for _ in gen:
        pass


class TPFusionEngine(FusionEngine):
    """FusionEngine with head / column sharded weights and all-reduce (see the module docstring).  Same op set, same reference
    op order; every rank returns the full noise prediction."""

    def __init__(self, cfg, get, ops, tp: TensorShard, heads_cfg=None, cache_step_invariants=False, precision="bf16",
                 fp8_attention=False):
        """precision="fp8" (round 6; BASELINE config 5 names TP = 8): the DiT blocks' linears through the reference's fp8 linear
        (vram_management/layers.py:115-151), sharded like the bf16 ones.  Column-parallel linears (q|k|v, FFN-1, cross-attention q /
        k|v) see the full K row: each rank quantises the replicated activation exactly as the unsharded engine does and computes its
        output columns bit for bit.  Row-parallel linears (o, cross-attention o, FFN-2) hold a K-slice of the activation, but the
        per-row scale_a of layers.py:126-133 is a property of the FULL row: the local row maxima are all-reduced (MAX, fp32 [rows]:
        131 KB at L = 32 760) BEFORE quantising, so every rank divides by the same scale (fw_row_absmax, fw_fp8_quant_rows_amax);
        the partial products (x scale_a, no bias) are all-reduced like the bf16 ones and the e4m3-rounded bias rides in the
        epilogue after the reduction.  fp8_attention: this rank's heads through fw_attention_fp8."""
        if precision not in ("bf16", "fp8"):
            raise ValueError(f"precision must be 'bf16' or 'fp8', got {precision!r}")
        for name, H in (("DiT", cfg.num_heads), ("VGGT", cfg.vggt_heads)):
            if not tp.divides(H):
                raise ValueError(f"{H} {name} heads do not divide over {tp.world} tensor-parallel ranks")
        self.tp = tp
        # (fp8_attention = "bicross" / "all" are one-GPU experiments of the unsharded engine: here, as under the sequence shard, they mean
        #  True -- the DiT self-attention of this rank's heads)
        super().__init__(cfg, get, ops, shard=None, heads_cfg=heads_cfg, cache_step_invariants=cache_step_invariants,
                         precision=precision, fp8_attention=bool(fp8_attention))

    # ------------------------------------------------------------------------------------------------ packing (weight slices)
    def _pack_dit(self, b, g, lin, lin_cat, prefix=None, adapter=None):
        cfg, ops, tp = self.cfg, self.ops, self.tp
        p = cfg.dit_prefix(b) if prefix is None else prefix
        hd = cfg.head_dim
        h0, h1 = tp.heads(cfg.num_heads)
        c0, c1 = h0 * hd, h1 * hd
        f8 = self.precision == "fp8"           # the DiT block's nn.Linear modules (engine.py _pack_dit): q/k/v/o of both attentions, FFN
        pk = lambda w, bias=None: ops.pack_linear(w, bias)
        pk8 = (lambda w, bias=None: ops.pack_linear(w, bias, fp8=True)) if f8 else pk
        # a row-parallel linear's bias is applied AFTER the reduction (fw_residual_add): under fp8 it is the value the fp8 linear
        # adds -- rounded to bf16 and, inside AutoWrappedLinear, through e4m3 (layers.py:138,158-159)
        rbias = (lambda n: ops.fp8_bias(g(n))) if f8 else (lambda n: ops.to_f32(g(n)))
        rows = lambda n: (g(n + ".weight")[c0:c1], g(n + ".bias")[c0:c1])
        cat_rows = lambda names: pk8(torch.cat([rows(n)[0] for n in names], 0), torch.cat([rows(n)[1] for n in names], 0))
        kslice = lambda n: pk8(g(n + ".weight")[:, c0:c1].contiguous())                 # row-parallel: bias after the reduce
        blk = _DitBlock()
        blk.index = b
        blk.mod = ops.to_f32(g(p + "modulation").reshape(6, cfg.dim))
        blk.qkv = cat_rows([p + "self_attn.q", p + "self_attn.k", p + "self_attn.v"])
        blk.o, blk.o_b = kslice(p + "self_attn.o"), rbias(p + "self_attn.o.bias")
        blk.norm_q = ops.to_f32(g(p + "self_attn.norm_q.weight")[c0:c1])
        blk.norm_k = ops.to_f32(g(p + "self_attn.norm_k.weight")[c0:c1])
        blk.cq = cat_rows([p + "cross_attn.q"])
        blk.ckv = cat_rows([p + "cross_attn.k", p + "cross_attn.v"])
        blk.co, blk.co_b = kslice(p + "cross_attn.o"), rbias(p + "cross_attn.o.bias")
        blk.cnorm_q = ops.to_f32(g(p + "cross_attn.norm_q.weight")[c0:c1])
        blk.cnorm_k = ops.to_f32(g(p + "cross_attn.norm_k.weight")[c0:c1])
        if cfg.has_image_input:
            blk.ckv_img = cat_rows([p + "cross_attn.k_img", p + "cross_attn.v_img"])
            blk.cnorm_k_img = ops.to_f32(g(p + "cross_attn.norm_k_img.weight")[c0:c1])
        blk.adapter = cfg.has_adapter(b) if adapter is None else bool(adapter)
        if blk.adapter:
            a = p + "cross_attn.processor."
            rp = _ru64(cfg.adapter_reduced)
            blk.a_g1 = lin(a + "k_proj.group1.weight", a + "k_proj.group1.bias")                          # replicated
            # 5120 -> 1024 row-parallel over this rank's head columns; its bias rides on rank 0's partial sums (the ReLU follows
            # the reduction)
            blk.a_g20 = pk(g(a + "k_proj.group2.0.weight")[:, c0:c1].contiguous(),
                           g(a + "k_proj.group2.0.bias") if tp.rank == 0 else None)
            blk.a_g22 = lin(a + "k_proj.group2.2.weight", a + "k_proj.group2.2.bias")
            blk.a_v0 = lin(a + "v_proj.group2.0.weight", a + "v_proj.group2.0.bias", n_pad=rp)
            blk.a_v2 = pk(_pad_to(g(a + "v_proj.group2.2.weight")[c0:c1], 1, rp), g(a + "v_proj.group2.2.bias")[c0:c1])   # column-parallel
        blk.norm3_w = ops.to_f32(g(p + "norm3.weight"))
        blk.norm3_b = ops.to_f32(g(p + "norm3.bias"))
        f0, f1 = tp.units(cfg.ffn_dim)
        blk.ffn0 = pk8(g(p + "ffn.0.weight")[f0:f1], g(p + "ffn.0.bias")[f0:f1])
        blk.ffn2, blk.ffn2_b = pk8(g(p + "ffn.2.weight")[:, f0:f1].contiguous()), rbias(p + "ffn.2.bias")
        return blk

    def _pack_vggt(self, p, g, lin):
        cfg, ops, tp = self.cfg, self.ops, self.tp
        C = cfg.vggt_dim
        hd = C // cfg.vggt_heads
        h0, h1 = tp.heads(cfg.vggt_heads)
        c0, c1 = h0 * hd, h1 * hd
        blk = _VggtBlock()
        blk.mod = ops.to_f32(g(p + "modulation").reshape(6, C))
        blk.norm1 = (ops.to_f32(g(p + "norm1.weight")), ops.to_f32(g(p + "norm1.bias")))
        w, bias = g(p + "attn.qkv.weight"), g(p + "attn.qkv.bias")                       # rows: q | k | v blocks of C
        sel = torch.cat([torch.arange(c0, c1) + i * C for i in range(3)])
        blk.qkv = ops.pack_linear(w[sel], bias[sel])
        blk.q_norm = (ops.to_f32(g(p + "attn.q_norm.weight")), ops.to_f32(g(p + "attn.q_norm.bias")))
        blk.k_norm = (ops.to_f32(g(p + "attn.k_norm.weight")), ops.to_f32(g(p + "attn.k_norm.bias")))
        blk.proj, blk.proj_b = ops.pack_linear(g(p + "attn.proj.weight")[:, c0:c1].contiguous(), None), ops.to_f32(g(p + "attn.proj.bias"))
        blk.ls1 = ops.to_f32(g(p + "ls1.gamma"))
        blk.norm2 = (ops.to_f32(g(p + "norm2.weight")), ops.to_f32(g(p + "norm2.bias")))
        m0, m1 = tp.units(cfg.vggt_mlp)
        blk.fc1 = ops.pack_linear(g(p + "mlp.fc1.weight")[m0:m1], g(p + "mlp.fc1.bias")[m0:m1])
        blk.fc2, blk.fc2_b = ops.pack_linear(g(p + "mlp.fc2.weight")[:, m0:m1].contiguous(), None), ops.to_f32(g(p + "mlp.fc2.bias"))
        blk.ls2 = ops.to_f32(g(p + "ls2.gamma"))
        return blk

    def _pack_bicross(self, p, g, lin, lin_cat):
        cfg, ops, tp = self.cfg, self.ops, self.tp
        bc = _Bicross()
        c = p + "cross_attn."
        bc.gamma1 = ops.to_f32(g(p + "gamma_m1"))
        bc.gamma2 = ops.to_f32(g(p + "gamma_m2"))
        bc.by_heads = tp.divides(cfg.bicross_heads)
        if bc.by_heads:
            hd = cfg.bicross_dim // cfg.bicross_heads
            h0, h1 = tp.heads(cfg.bicross_heads)
            c0, c1 = h0 * hd, h1 * hd
            two = lambda a, b: ops.pack_linear(torch.cat([g(a + ".weight")[c0:c1], g(b + ".weight")[c0:c1]], 0),
                                               torch.cat([g(a + ".bias")[c0:c1], g(b + ".bias")[c0:c1]], 0))
            bc.qv1 = two(c + "m1_proj", c + "values_m1_proj")
            bc.kv2 = two(c + "m2_proj", c + "values_m2_proj")
            # K = (12 / n) * 96 is 64-aligned only for some n: pad the slab (zeros) and the activation columns alike
            bc.kpad = _ru64(c1 - c0)
            bc.out1 = ops.pack_linear(_pad_to(g(c + "out_m1_proj.weight")[:, c0:c1], 1, bc.kpad), None)
            bc.out2 = ops.pack_linear(_pad_to(g(c + "out_m2_proj.weight")[:, c0:c1], 1, bc.kpad), None)
            bc.out1_b, bc.out2_b = ops.to_f32(g(c + "out_m1_proj.bias")), ops.to_f32(g(c + "out_m2_proj.bias"))
        else:                   # 12 heads, 8 ranks: projections replicated, attention split by query rows
            bc.qv1 = lin_cat([c + "m1_proj", c + "values_m1_proj"])
            bc.kv2 = lin_cat([c + "m2_proj", c + "values_m2_proj"])
            bc.out1 = lin(c + "out_m1_proj.weight", c + "out_m1_proj.bias")
            bc.out2 = lin(c + "out_m2_proj.weight", c + "out_m2_proj.bias")
        return bc

    # ------------------------------------------------------------------------------------------------ helpers
    def _reduce_into(self, stream, x_in, lin, bias, g1=None, g0=None, kind="all_reduce"):
        """stream += (all_reduce(x_in @ lin^T) + bias) * g1 + g0, as a generator: row blocks so that the reduction of block c flies
        behind the GEMM of block c+1 (and behind whatever the interleaved branch enqueues at the yield)."""
        ops, tp = self.ops, self.tp
        M = x_in.shape[0]
        nch = max(1, min(tp.chunks, M // tp.chunk_rows)) if tp.world > 1 else 1
        cuts = [M * i // nch // 8 * 8 for i in range(nch)] + [M]
        pend = []
        f32 = tp.reduce_dtype == torch.float32
        rows_of = lambda a, b: x_in[a:b]
        if lin.fp8:
            # x_in is this rank's K-slice; the fp8 linear's per-row scale belongs to the FULL row: MAX of the local maxima over the
            # group, then every rank quantises its slice with the same scale (bit-identical to slicing the unsharded quantised row)
            p_amax = tp.all_reduce_async(ops.row_absmax(x_in), "all_reduce_amax", op=dist.ReduceOp.MAX)
            yield
            q8, scale = ops.quantize_fp8_rows(x_in, amax=p_amax.wait())
            rows_of = lambda a, b: (q8[a:b], scale[a:b])
        for a, b in zip(cuts[:-1], cuts[1:]):
            part = ops.linear(rows_of(a, b), lin, out_f32=f32)
            pend.append((a, b, tp.all_reduce_async(part, kind)))
        yield
        for a, b, p in pend:
            ops.residual_add(stream[a:b], p.wait(), bias=bias, g1=g1, g0=g0)

    def _stats(self, *slices):
        """Sum of squares over the FULL width of each row for several (rows, local-columns) slices: one fused all-reduce."""
        ops, tp = self.ops, self.tp
        buf = torch.empty(sum(s.shape[0] for s in slices), dtype=torch.float32, device=slices[0].device)
        outs, a = [], 0
        for s in slices:
            outs.append(ops.row_sumsq(s, out=buf[a:a + s.shape[0]]))
            a += s.shape[0]
        return tp.all_reduce_async(buf, "all_reduce_stats"), outs

    # ------------------------------------------------------------------------------------------------ blocks (generators)
    def _tp_dit_attn(self, blk, x, ctx_txt, ctx_img, t_mod, tabs, plucker, out):
        """Self-attention + cross-attention (+ camera adapter) of one DiT block; out["mod"] receives the modulation table."""
        cfg, ops, tp = self.cfg, self.ops, self.tp
        D, hd = cfg.dim, cfg.head_dim
        Hl = cfg.num_heads // tp.world
        W = Hl * hd
        h0 = tp.heads(cfg.num_heads)[0]
        mod = out["mod"] = self._dit_mod(blk, t_mod)
        xn = ops.layernorm(x, scale=mod[1], shift=mod[0], eps=cfg.eps)
        qkv = ops.linear(xn, blk.qkv)                                   # [L, 3 W]: this rank's heads
        q, k, v = qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:]
        pend, (sq, sk) = self._stats(q, k)
        yield
        pend.wait()
        tab = tabs["dit"]
        if self.fp8_attention:       # this rank's heads on e4m3 q | k | v (fw_attention_fp8): q / k written as e4m3 by the q/k pass
            qk8 = ops.empty(q.shape[0], 2 * W, dtype=torch.uint8)
            ops.qk_prep(q, Hl, hd, norm="rms_full", norm_w=blk.norm_q, eps=cfg.eps, rope="interleaved", table=tab,
                        out_scale=ops.q_scale_fp8(hd), ext_sumsq=sq, norm_width=D, out8=qk8[:, :W])
            ops.qk_prep(k, Hl, hd, norm="rms_full", norm_w=blk.norm_k, eps=cfg.eps, rope="interleaved", table=tab,
                        ext_sumsq=sk, norm_width=D, out8=qk8[:, W:])
            vt8, Lk = ops.prepare_v_fp8(v, Hl, hd)
            o = ops.attention_fp8(qk8[:, :W], qk8[:, W:], vt8, Hl, hd, Lk)
        else:
            ops.qk_prep(q, Hl, hd, norm="rms_full", norm_w=blk.norm_q, eps=cfg.eps, rope="interleaved", table=tab,
                        out_scale=ops.q_scale(hd), ext_sumsq=sq, norm_width=D)
            ops.qk_prep(k, Hl, hd, norm="rms_full", norm_w=blk.norm_k, eps=cfg.eps, rope="interleaved", table=tab,
                        ext_sumsq=sk, norm_width=D)
            o = ops.attention(q, k, v, Hl, hd, batch=1, q_prescaled=True)
        yield from self._reduce_into(x, o, blk.o, blk.o_b, g1=mod[2])
        # cross-attention: text + image keys share q (wan_video_dit.py:185-201)
        xn3 = ops.layernorm(x, w=blk.norm3_w, b=blk.norm3_b, eps=cfg.eps)
        qc = ops.linear(xn3, blk.cq)
        kv = ops.linear(ctx_txt, blk.ckv)
        sl = [qc, kv[:, :W]]
        kvi = None
        if ctx_img is not None:
            kvi = ops.linear(ctx_img, blk.ckv_img)
            sl.append(kvi[:, :W])
        pend, ss = self._stats(*sl)
        yield
        pend.wait()
        ops.qk_prep(qc, Hl, hd, norm="rms_full", norm_w=blk.cnorm_q, eps=cfg.eps, out_scale=ops.q_scale(hd), ext_sumsq=ss[0], norm_width=D)
        ops.qk_prep(kv[:, :W], Hl, hd, norm="rms_full", norm_w=blk.cnorm_k, eps=cfg.eps, ext_sumsq=ss[1], norm_width=D)
        oc = ops.attention(qc, kv[:, :W], kv[:, W:], Hl, hd, batch=1, q_prescaled=True)
        if kvi is not None:
            ops.qk_prep(kvi[:, :W], Hl, hd, norm="rms_full", norm_w=blk.cnorm_k_img, eps=cfg.eps, ext_sumsq=ss[2], norm_width=D)
            ops.attention(qc, kvi[:, :W], kvi[:, W:], Hl, hd, batch=1, out=oc, accumulate=True, q_prescaled=True)
        if blk.adapter and plucker is not None:
            # camera_control.py:109-127 ('adaln', scale == 0): x <- x + shift, shift computed from ALL channels of x
            t1 = ops.linear(oc, blk.a_g20, out_f32=tp.reduce_dtype == torch.float32)       # partial sums over my channels
            p1 = tp.all_reduce_async(t1, "all_reduce_adapter")
            pterm = ops.linear(plucker, blk.a_g1)
            yield
            t1 = ops.activation(ops.cast_act(p1.wait()) if t1.dtype == torch.float32 else p1.wait(), "relu")
            comb = ops.linear(t1, blk.a_g22, res=pterm, out=pterm)
            t2 = ops.linear(comb, blk.a_v0, act="relu")
            ops.linear(t2, blk.a_v2, res=oc, out=oc)                    # my columns of the shift
        yield from self._reduce_into(x, oc, blk.co, blk.co_b)

    def _tp_dit_ffn(self, blk, x, mod):
        cfg, ops = self.cfg, self.ops
        xn = ops.layernorm(x, scale=mod[4], shift=mod[3], eps=cfg.eps)
        h = ops.linear(xn, blk.ffn0, act="gelu_tanh")                   # [L, Fd / n]
        yield from self._reduce_into(x, h, blk.ffn2, blk.ffn2_b, g1=mod[5])

    def _tp_vggt_attn(self, blk, tok, e0, tabs, batch, out):
        cfg, ops, tp = self.cfg, self.ops, self.tp
        C = cfg.vggt_dim
        hd = C // cfg.vggt_heads
        Hl = cfg.vggt_heads // tp.world
        W = Hl * hd
        e = out["e"] = self._vggt_mod(blk, e0)
        xn = ops.layernorm(tok, w=blk.norm1[0], b=blk.norm1[1], scale=e[1], shift=e[0], eps=cfg.vggt_eps)
        qkv = ops.linear(xn, blk.qkv)
        q, k, v = qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:]
        ops.qk_prep(q, Hl, hd, norm="ln_head", norm_w=blk.q_norm[0], norm_b=blk.q_norm[1], eps=cfg.vggt_eps, rope="half2d",
                    table=tabs["vggt"], out_scale=ops.q_scale(hd))
        ops.qk_prep(k, Hl, hd, norm="ln_head", norm_w=blk.k_norm[0], norm_b=blk.k_norm[1], eps=cfg.vggt_eps, rope="half2d",
                    table=tabs["vggt"])
        o = ops.attention(q, k, v, Hl, hd, batch=batch, q_prescaled=True)
        yield from self._reduce_into(tok, o, blk.proj, blk.proj_b, g1=blk.ls1)

    def _tp_vggt_mlp(self, blk, tok, e):
        cfg, ops = self.cfg, self.ops
        xn = ops.layernorm(tok, w=blk.norm2[0], b=blk.norm2[1], eps=cfg.vggt_eps)
        h = ops.linear(xn, blk.fc1, act="gelu_erf")
        g1, g0 = self._vggt_gates(blk, e)
        yield from self._reduce_into(tok, h, blk.fc2, blk.fc2_b, g1=g1, g0=g0)

    def _tp_bicross(self, bc, x, tok, tabs):
        cfg, ops, tp = self.cfg, self.ops, self.tp
        Bd, Hb = cfg.bicross_dim, cfg.bicross_heads
        hd = Bd // Hb
        a = ops.layernorm(x, eps=1e-6)
        b = ops.layernorm(tok, eps=1e-6)
        qv1 = ops.linear(a, bc.qv1)
        kv2 = ops.linear(b, bc.kv2)
        if bc.by_heads:
            Hl = Hb // tp.world
            W = Hl * hd
            ops.qk_prep(qv1[:, :W], Hl, hd, rope="interleaved", table=tabs["bi_dit"], out_scale=ops.q_scale(hd))
            ops.qk_prep(kv2[:, :W], Hl, hd, rope="interleaved", table=tabs["bi_agg"])
            L, L2 = qv1.shape[0], kv2.shape[0]
            # attention outputs land in 64-aligned (zero-padded) slabs so that they are the K operand of the sliced out-projections
            o1 = torch.zeros(L, bc.kpad, dtype=qv1.dtype, device=qv1.device) if bc.kpad != W else None
            o2 = torch.zeros(L2, bc.kpad, dtype=qv1.dtype, device=qv1.device) if bc.kpad != W else None
            r1 = ops.attention(qv1[:, :W], kv2[:, :W], kv2[:, W:], Hl, hd, batch=1, q_prescaled=True, out=None if o1 is None else o1[:, :W])
            r2 = ops.attention(kv2[:, :W], qv1[:, :W], qv1[:, W:], Hl, hd, batch=1, q_prescaled=True, out=None if o2 is None else o2[:, :W])
            o1 = r1 if o1 is None else o1
            o2 = r2 if o2 is None else o2
            yield from self._reduce_into(x, o1, bc.out1, bc.out1_b, g1=bc.gamma1)
            yield from self._reduce_into(tok, o2, bc.out2, bc.out2_b, g1=bc.gamma2)
            return
        # query-row split: every rank holds all of q|v1 and k|v2 (replicated projections) and attends for its rows only
        ops.qk_prep(qv1[:, :Bd], Hb, hd, rope="interleaved", table=tabs["bi_dit"], out_scale=ops.q_scale(hd))
        ops.qk_prep(kv2[:, :Bd], Hb, hd, rope="interleaved", table=tabs["bi_agg"])
        a1, b1, c1 = tp.rows(qv1.shape[0])
        a2, b2, c2 = tp.rows(kv2.shape[0])
        o1 = ops.attention(qv1[a1:b1, :Bd], kv2[:, :Bd], kv2[:, Bd:], Hb, hd, batch=1, q_prescaled=True)
        u1 = ops.linear(o1, bc.out1, g1=bc.gamma1)                      # gamma1 * (W o1 + b) for my rows
        p1 = tp.all_gather_rows_async(u1, c1)
        o2 = ops.attention(kv2[a2:b2, :Bd], qv1[:, :Bd], qv1[:, Bd:], Hb, hd, batch=1, q_prescaled=True)
        u2 = ops.linear(o2, bc.out2, g1=bc.gamma2)
        p2 = tp.all_gather_rows_async(u2, c2)
        yield
        ops.residual_add(x, p1.wait())
        ops.residual_add(tok, p2.wait())

    # ------------------------------------------------------------------------------------------------ forward
    def joint_forward_pair(self, *a, **k):
        raise NotImplementedError("merged CFG is the single-GPU form; under TP the two passes go to two rank groups")

    def _forward(self, x, timestep, contexts, clip_feature, y, plucker_fea, plucker_context_lens, uncond, return_prediction,
                 camera_token, control_camera_latents_input, collect):
        cfg = self.cfg
        assert len(contexts) == 1 and x.shape[0] == 1 and self.shard is None
        # embeddings, patchify, bridge and head are replicated: the unsharded engine's own code (engine.py _prologue ...)
        st = self._prologue(x, timestep, contexts, clip_feature, y, plucker_fea, control_camera_latents_input)
        F, P, tabs, xs, t_mod, e0 = st.F, st.P, st.tabs, st.xs, st.t_mod, st.e0
        ctx_txt, ctx_img, plucker = st.ctx_txt, st.ctx_img, st.plucker
        per_block = None if collect is None else collect.get("per_block")
        self._build_modulation(t_mod, e0)

        for b in range(cfg.start_index):
            blk, sd = self.dit[b], {}
            _drain(self._tp_dit_attn(blk, xs, ctx_txt, ctx_img, t_mod, tabs, plucker, sd))
            _drain(self._tp_dit_ffn(blk, xs, sd["mod"]))
            if per_block is not None:
                per_block("x", b, xs)
        if collect is not None:
            collect["x_after_pcb"] = xs.clone()

        tok, _ = self._entry_tokens(st, camera_token)
        if collect is not None:
            collect["tokens_in"] = tok.clone()

        need = self._layers_for_heads(return_prediction)
        outputs = {}
        for i in range(cfg.n_irg):
            fb, blk, gb = self.frame[i], self.dit[cfg.start_index + i], self.glob[i]
            sd, sf, sg = {}, {}, {}

            def vggt_branch():
                # frame block (attention + MLP), then the global block's attention half (aggregator.py:215-260, block.py:59-69)
                yield from self._tp_vggt_attn(fb, tok, e0, tabs, F, sf)
                yield from self._tp_vggt_mlp(fb, tok, sf["e"])
                if i in need:
                    sf["frame_out"] = tok.clone()
                yield from self._tp_vggt_attn(gb, tok, e0, tabs, 1, sg)

            # the DiT partial and the VGGT partials are independent until the bicross: their collectives hide behind each other
            _interleave(self._tp_dit_attn(blk, xs, ctx_txt, ctx_img, t_mod, tabs, plucker, sd), vggt_branch())
            if i in cfg.cross_attention_list and not uncond:
                _drain(self._tp_bicross(self.bicross[cfg.cross_attention_list.index(i)], xs, tok, tabs))
            _interleave(self._tp_dit_ffn(blk, xs, sd["mod"]), self._tp_vggt_mlp(gb, tok, sg["e"]))
            if per_block is not None:
                per_block("x", cfg.start_index + i, xs)
                per_block("tok", i, tok)
            if i in need:
                outputs[i] = torch.cat([sf["frame_out"].view(1, F, P, -1), tok.view(1, F, P, -1)], dim=-1)
        if collect is not None:
            collect["x_final"] = xs.clone()
            collect["tokens_final"] = tok.clone()
        return self._epilogue(st, x.dtype, outputs, return_prediction)
