"""TEST INFRASTRUCTURE ONLY -- CPU restatement (PyTorch fp32) of FantasyWorld's joint_forward.

This is the oracle the HIP path is checked against.  It is a functional, module-free restatement of the reference
algorithm; every function cites the reference lines it follows (paths relative to the reference repo root).  It is
pinned in two ways (see oracle/make_golden.py and tests/test_oracle_pin.py):
  * in the build container it is compared against the REAL reference modules imported from /root/reference
    (oracle/ref_harness.py) on seeded weights/inputs;
  * the outputs of the real reference on those inputs are committed under tests/golden/*.pt and the oracle must
    reproduce them on any machine (the GPU box has no /root/reference).
The reference has no golden vectors or tests of its own for this path: the pin is "reference code
executed on CPU in fp32 with deterministic synthetic weights".

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------------------
# embeddings / rotary tables
# ---------------------------------------------------------------------------------------------------------------
def sinusoidal_embedding_1d(dim, position):
    # FantasyWorld/diffsynth_wan21/models/wan_video_dit.py:73-77
    sinusoid = torch.outer(position.type(torch.float64),
                           torch.pow(10000, -torch.arange(dim // 2, dtype=torch.float64).div(dim // 2)))
    x = torch.cat([torch.cos(sinusoid), torch.sin(sinusoid)], dim=1)
    return x.to(position.dtype)


def precompute_freqs_cis(dim, end=1024, theta=10000.0):
    # wan_video_dit.py:88-94
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].double() / dim))
    freqs = torch.outer(torch.arange(end), freqs)
    return torch.polar(torch.ones_like(freqs), freqs)


def precompute_freqs_cis_3d(dim, end=1024, theta=10000.0):
    # wan_video_dit.py:80-85
    return (precompute_freqs_cis(dim - 2 * (dim // 3), end, theta),
            precompute_freqs_cis(dim // 3, end, theta),
            precompute_freqs_cis(dim // 3, end, theta))


def expand_freqs(freqs3, f, h, w):
    # FantasyWorld/fusion/model_wan21.py:132-136
    return torch.cat([
        freqs3[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1),
        freqs3[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
        freqs3[2][:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(f * h * w, 1, -1)


def build_freqs_3d_with_extra_cis(freqs3, f, h, w, n_extra):
    # wan_video_dit.py:105-132: n_extra un-rotated (1+0j) rows in front of every frame
    patch = expand_freqs(freqs3, f, h, w).reshape(f, h * w, -1)
    extra = torch.ones(f, n_extra, patch.size(-1), dtype=patch.dtype)
    return torch.cat([extra, patch], dim=1).reshape(f * (n_extra + h * w), 1, -1)


def rope_apply(x, freqs, num_heads):
    # wan_video_dit.py:97-102: complex multiply on interleaved pairs, in fp64
    b, s, _ = x.shape
    xh = x.view(b, s, num_heads, -1)
    xc = torch.view_as_complex(xh.to(torch.float64).reshape(b, s, num_heads, -1, 2))
    return torch.view_as_real(xc * freqs).flatten(2).to(x.dtype)


def rms_norm(x, weight, eps):
    # wan_video_dit.py:135-146
    return (x.float() * torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + eps)).to(x.dtype) * weight


def sdpa(q, k, v, num_heads):
    # wan_video_dit.py:59-65 (no flash-attn installed -> F.scaled_dot_product_attention), written out explicitly
    b, lq, _ = q.shape
    lk = k.shape[1]
    qh = q.view(b, lq, num_heads, -1).transpose(1, 2)
    kh = k.view(b, lk, num_heads, -1).transpose(1, 2)
    vh = v.view(b, lk, num_heads, -1).transpose(1, 2)
    if b * num_heads * lq * lk <= (1 << 28):
        s = qh @ kh.transpose(-1, -2) / math.sqrt(qh.shape[-1])
        o = torch.softmax(s, dim=-1) @ vh
    else:   # the score tensor would not fit the host memory: the call the reference itself makes (wan_video_dit.py:62)
        o = F.scaled_dot_product_attention(qh, kh, vh)
    return o.transpose(1, 2).reshape(b, lq, -1)


# fp8 mode (BASELINE config 5): the nn.Linear modules of the DiT blocks computed by AutoWrappedLinear.fp8_linear, as
# enable_vram_management(dit, module_map={nn.Linear: AutoWrappedLinear}, computation_dtype=float8_e4m3fn) would make them
# (diffsynth_wan21/vram_management/layers.py:145-166; diffsynth_wan22/vram_management/layers.py:113-151).  Set through
# joint_forward(fp8_linears=True); the camera-adapter processor's small linears keep full precision (the engine's choice, stated in
# docs/parity.md -- the reference defines no fp8 run of the fusion model at all).
_FP8 = {"on": False}
_FP8_SITES = (".self_attn.q", ".self_attn.k", ".self_attn.v", ".self_attn.o", ".cross_attn.q", ".cross_attn.k", ".cross_attn.v",
              ".cross_attn.o", ".cross_attn.k_img", ".cross_attn.v_img", ".ffn.0", ".ffn.2")


def fp8_linear(x, weight, bias):
    """AutoWrappedLinear.fp8_linear (diffsynth_wan22/vram_management/layers.py:115-151) for e4m3fn, with torch._scaled_mm written
    as its definition (a @ b) * scale_a * scale_b + bias -> out_dtype (the call itself only exists on GPU back-ends).  The
    reference runs it on bf16 activations / weights and gets bf16 back: the fp32 pipeline of this oracle rounds at those two points."""
    shape = x.shape
    xb = x.reshape(-1, shape[-1]).to(torch.bfloat16)                                    # origin_dtype = bf16
    x_max = torch.max(torch.abs(xb), dim=-1, keepdim=True).values
    scale_a = torch.clamp(x_max / 448.0, min=1.0).float()
    xq = (xb / (scale_a + 1e-8)).to(torch.float8_e4m3fn)
    wq = weight.to(torch.bfloat16).to(torch.float8_e4m3fn)
    # AutoWrappedLinear.forward hands fp8_linear weight AND bias already cast to the computation dtype (layers.py:158-159)
    bias = bias.to(torch.bfloat16).to(torch.float8_e4m3fn)
    y = (xq.float() @ wq.float().t()) * scale_a + bias.to(torch.bfloat16).float()
    return y.to(torch.bfloat16).float().reshape(*shape[:-1], -1)


def linear(x, W, name):
    if _FP8["on"] and name.endswith(_FP8_SITES) and ("pipe.dit.blocks." in name or ".x_dit." in name):
        return fp8_linear(x, W[name + ".weight"], W[name + ".bias"])
    return F.linear(x, W[name + ".weight"], W[name + ".bias"])


# ---------------------------------------------------------------------------------------------------------------
# DiT block (wan_video_dit.py:254-321), attention processors (wan_video_dit.py:185-201, camera_control.py:92-148)
# ---------------------------------------------------------------------------------------------------------------
def dit_self_attn(x, W, p, freqs, cfg):
    # wan_video_dit.py:175-182
    q = rms_norm(linear(x, W, p + "q"), W[p + "norm_q.weight"], cfg.eps)
    k = rms_norm(linear(x, W, p + "k"), W[p + "norm_k.weight"], cfg.eps)
    v = linear(x, W, p + "v")
    q = rope_apply(q, freqs, cfg.num_heads)
    k = rope_apply(k, freqs, cfg.num_heads)
    return linear(sdpa(q, k, v, cfg.num_heads), W, p + "o")


def dit_cross_attn(x, ctx_all, W, p, cfg, adapter, plucker_fea):
    # wan_video_dit.py:185-201 ; with the 'adaln' adapter camera_control.py:92-148
    if cfg.has_image_input:
        img, ctx = ctx_all[:, :cfg.clip_tokens], ctx_all[:, cfg.clip_tokens:]
    else:
        ctx = ctx_all
    q = rms_norm(linear(x, W, p + "q"), W[p + "norm_q.weight"], cfg.eps)
    k = rms_norm(linear(ctx, W, p + "k"), W[p + "norm_k.weight"], cfg.eps)
    v = linear(ctx, W, p + "v")
    o = sdpa(q, k, v, cfg.num_heads)
    if cfg.has_image_input:
        k_img = rms_norm(linear(img, W, p + "k_img"), W[p + "norm_k_img.weight"], cfg.eps)
        v_img = linear(img, W, p + "v_img")
        o = o + sdpa(q, k_img, v_img, cfg.num_heads)
    if adapter and plucker_fea is not None:
        a = p + "processor."
        is_all_zeros = bool(torch.all(plucker_fea == 0).item())                       # camera_control.py:111
        out1 = linear(plucker_fea, W, a + "k_proj.group1")                            # GroupLinearDualK :36-39
        out2 = linear(F.relu(linear(o, W, a + "k_proj.group2.0")), W, a + "k_proj.group2.2")
        combined = out2 + out1                                                        # :116
        scale = 0.0                                                                   # GroupLinearDualV :59-63
        shift = linear(F.relu(linear(combined, W, a + "v_proj.group2.0")), W, a + "v_proj.group2.2")
        if not is_all_zeros:
            o = o * (scale + 1.0) + shift                                             # :124-127
    return linear(o, W, p + "o")


def dit_block_partial(x, ctx_all, t_mod, freqs, W, p, cfg, adapter, plucker_fea):
    # wan_video_dit.py:296-306
    mods = (W[p + "modulation"] + t_mod).chunk(6, dim=1)
    shift_msa, scale_msa, gate_msa = mods[0], mods[1], mods[2]
    D = x.shape[-1]
    inp = F.layer_norm(x, (D,), None, None, cfg.eps) * (1 + scale_msa) + shift_msa
    x = x + gate_msa * dit_self_attn(inp, W, p + "self_attn.", freqs, cfg)
    n3 = F.layer_norm(x, (D,), W[p + "norm3.weight"], W[p + "norm3.bias"], cfg.eps)
    x = x + dit_cross_attn(n3, ctx_all, W, p + "cross_attn.", cfg, adapter, plucker_fea)
    return x, (mods[3], mods[4], mods[5])


def dit_block_remaining(x, mods, W, p, cfg):
    # wan_video_dit.py:288-294
    shift_mlp, scale_mlp, gate_mlp = mods
    D = x.shape[-1]
    inp = F.layer_norm(x, (D,), None, None, cfg.eps) * (1 + scale_mlp) + shift_mlp
    hdn = F.gelu(linear(inp, W, p + "ffn.0"), approximate="tanh")
    return x + gate_mlp * linear(hdn, W, p + "ffn.2")


# ---------------------------------------------------------------------------------------------------------------
# VGGT block (vggt/layers/block.py:73-116), attention (attention.py:50-72), 2-D RoPE (rope.py:82-188)
# ---------------------------------------------------------------------------------------------------------------
def rope2d(tokens, positions, base):
    # tokens [B, H, N, hd], positions [B, N, 2]
    hd = tokens.shape[-1]
    fd = hd // 2
    exponents = torch.arange(0, fd, 2).float() / fd
    inv_freq = 1.0 / (base ** exponents)
    max_pos = int(positions.max()) + 1
    angles = torch.einsum("i,j->ij", torch.arange(max_pos, dtype=inv_freq.dtype), inv_freq)
    angles = torch.cat((angles, angles), dim=-1)
    cos_c, sin_c = angles.cos(), angles.sin()

    def rot(x):
        x1, x2 = x[..., : fd // 2], x[..., fd // 2:]
        return torch.cat((-x2, x1), dim=-1)

    def apply(t, pos):
        cos = F.embedding(pos, cos_c)[:, None, :, :]
        sin = F.embedding(pos, sin_c)[:, None, :, :]
        return t * cos + rot(t) * sin

    vert, horiz = tokens.chunk(2, dim=-1)
    return torch.cat((apply(vert, positions[..., 0]), apply(horiz, positions[..., 1])), dim=-1)


def vggt_attention(x, pos, W, p, cfg):
    # attention.py:50-72
    B, N, C = x.shape
    H = cfg.vggt_heads
    hd = C // H
    qkv = linear(x, W, p + "qkv").reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    q = F.layer_norm(q, (hd,), W[p + "q_norm.weight"], W[p + "q_norm.bias"], cfg.vggt_eps)
    k = F.layer_norm(k, (hd,), W[p + "k_norm.weight"], W[p + "k_norm.bias"], cfg.vggt_eps)
    q = rope2d(q, pos, cfg.vggt_rope_freq)
    k = rope2d(k, pos, cfg.vggt_rope_freq)
    if B * H * N * N <= (1 << 28):
        s = q @ k.transpose(-1, -2) / math.sqrt(hd)
        o = torch.softmax(s, dim=-1) @ v
    else:   # as above: attention.py:61 calls F.scaled_dot_product_attention
        o = F.scaled_dot_product_attention(q, k, v)
    return linear(o.transpose(1, 2).reshape(B, N, C), W, p + "proj")


def vggt_block_partial(x, pos, e0, W, p, cfg):
    # block.py:91-107 ; e0 [1,6,C] is repeated over the batch of frames (:92-96)
    C = x.shape[-1]
    e = (W[p + "modulation"] + e0).chunk(6, dim=1)          # each [1,1,C]
    n1 = F.layer_norm(x, (C,), W[p + "norm1.weight"], W[p + "norm1.bias"], cfg.vggt_eps)
    x = x + W[p + "ls1.gamma"] * vggt_attention(n1 * (1 + e[1]) + e[0], pos, W, p + "attn.", cfg)   # :73-76
    return x, e


def vggt_block_remaining(x, e, W, p, cfg):
    # block.py:78-81: modulation applied AFTER the MLP, gate e[5] outside LayerScale, e[2] unused
    C = x.shape[-1]
    n2 = F.layer_norm(x, (C,), W[p + "norm2.weight"], W[p + "norm2.bias"], cfg.vggt_eps)
    m = linear(F.gelu(linear(n2, W, p + "mlp.fc1")), W, p + "mlp.fc2")
    return x + (W[p + "ls2.gamma"] * (m * (1 + e[4]) + e[3])) * e[5]


# ---------------------------------------------------------------------------------------------------------------
# bidirectional cross attention (fusion/layer/block.py:179-221, 532-625)
# ---------------------------------------------------------------------------------------------------------------
def bicross(x1, x2, freqs_dit, freqs_agg, W, p, cfg):
    H = cfg.bicross_heads
    a = F.layer_norm(x1, (x1.shape[-1],), None, None, 1e-6)
    b = F.layer_norm(x2, (x2.shape[-1],), None, None, 1e-6)
    c = p + "cross_attn."
    q = rope_apply(linear(a, W, c + "m1_proj"), freqs_dit, H)
    k = rope_apply(linear(b, W, c + "m2_proj"), freqs_agg, H)
    v1 = linear(a, W, c + "values_m1_proj")
    v2 = linear(b, W, c + "values_m2_proj")
    o1 = linear(sdpa(q, k, v2, H), W, c + "out_m1_proj")
    o2 = linear(sdpa(k, q, v1, H), W, c + "out_m2_proj")
    return x1 + W[p + "gamma_m1"] * o1, x2 + W[p + "gamma_m2"] * o2


# ---------------------------------------------------------------------------------------------------------------
# joint_forward (FantasyWorld/fusion/model_wan21.py:104-224)
# ---------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def control_adapter(ctl, W, pre):
    # FantasyWorld/diffsynth_wan22/models/wan_video_camera_controller.py:24-44 (SimpleAdapter.forward), :64-76 (ResidualBlock)
    bs, c, f, h, w = ctl.shape
    u = F.pixel_unshuffle(ctl.permute(0, 2, 1, 3, 4).reshape(bs * f, c, h, w), 8)
    xc = F.conv2d(u, W[pre + "conv.weight"], W[pre + "conv.bias"], stride=2)
    r = pre + "residual_blocks.0."
    out = F.conv2d(F.relu(F.conv2d(xc, W[r + "conv1.weight"], W[r + "conv1.bias"], padding=1)),
                   W[r + "conv2.weight"], W[r + "conv2.bias"], padding=1) + xc
    return out.view(bs, f, out.size(1), out.size(2), out.size(3)).permute(0, 2, 1, 3, 4)      # [b, D, f, h/16, w/16]


def joint_forward(W, cfg, x, timestep, context, clip_feature=None, y=None, plucker_fea=None,
                  plucker_context_lens=None, uncond=False, collect=None, control_camera_latents_input=None, fp8_linears=False,
                  camera_token=None):
    _FP8["on"] = bool(fp8_linears)
    try:
        return _joint_forward(W, cfg, x, timestep, context, clip_feature, y, plucker_fea, plucker_context_lens, uncond, collect,
                              control_camera_latents_input, camera_token)
    finally:
        _FP8["on"] = False


def _joint_forward(W, cfg, x, timestep, context, clip_feature=None, y=None, plucker_fea=None,
                   plucker_context_lens=None, uncond=False, collect=None, control_camera_latents_input=None, camera_token=None):
    """W: name -> fp32 tensor (reference parameter names). Returns noise_pred [1,16,F,H,W] (fp32).
    Pass collect={"output_list": {}} to receive the aggregator's output_list (layer -> [f, P, 2C]), the input of the geometry
    heads (oracle/fw_heads_oracle.py restates VGGT._head_predction on it)."""
    pd = "pipe.dit."
    f = x.shape[2]
    # A1 (wan_video_dit.py:393-399)
    t = linear(F.silu(linear(sinusoidal_embedding_1d(cfg.freq_dim, timestep), W, pd + "time_embedding.0")),
               W, pd + "time_embedding.2")
    t_mod = linear(F.silu(t), W, pd + "time_projection.1").unflatten(1, (6, cfg.dim))
    ctx = linear(F.gelu(linear(context, W, pd + "text_embedding.0"), approximate="tanh"), W, pd + "text_embedding.2")
    if cfg.control_adapter and y is not None:
        x = torch.cat([x, y], dim=1)                       # model_wan22.py:252-253 (require_vae_embedding)
    if cfg.has_image_input:
        x = torch.cat([x, y], dim=1)
        ie = pd + "img_emb.proj."
        c = F.layer_norm(clip_feature, (cfg.clip_dim,), W[ie + "0.weight"], W[ie + "0.bias"], 1e-5)
        c = linear(F.gelu(linear(c, W, ie + "1")), W, ie + "3")
        c = F.layer_norm(c, (cfg.dim,), W[ie + "4.weight"], W[ie + "4.bias"], 1e-5)
        ctx = torch.cat([c, ctx], dim=1)
    # patchify (wan_video_dit.py:424-435)
    x = F.conv3d(x, W[pd + "patch_embedding.weight"], W[pd + "patch_embedding.bias"], stride=(1, 2, 2))
    if cfg.control_adapter and control_camera_latents_input is not None:
        # diffsynth_wan22/models/wan_video_dit.py:390-396 (patchify: x + control_adapter(control), batch 1)
        x = x + control_adapter(control_camera_latents_input, W, pd + "control_adapter.")
    _, _, f, h, w = x.shape
    x = x.flatten(2).transpose(1, 2).contiguous()          # b (f h w) c
    hd = cfg.dim // cfg.num_heads
    freqs = expand_freqs(precompute_freqs_cis_3d(hd), f, h, w)
    fb = precompute_freqs_cis_3d(cfg.bicross_dim // cfg.bicross_heads)
    freqs_bi_dit = expand_freqs(fb, f, h, w)
    freqs_bi_agg = build_freqs_3d_with_extra_cis(fb, f, h, w, cfg.n_special)

    per_block = None if collect is None else collect.get("per_block")     # fn(kind, index, stream) after every block
    for b in range(cfg.start_index):
        p = cfg.dit_prefix(b)
        x, mods = dit_block_partial(x, ctx, t_mod, freqs, W, p, cfg, cfg.has_adapter(b), plucker_fea)
        x = dit_block_remaining(x, mods, W, p, cfg)
        if per_block is not None:
            per_block("x", b, x[0])
    if collect is not None:
        collect["x_after_pcb"] = x[0].clone()

    # bridge (model_wan21.py:170-175; vggt.py:118-131; aggregator.py:261-306)
    pt = F.linear(x, W["vggt.projection_head.weight"].reshape(cfg.vggt_dim, cfg.dim), W["vggt.projection_head.bias"])
    pt = pt.view(f, h * w, cfg.vggt_dim)
    e = linear(F.silu(linear(sinusoidal_embedding_1d(cfg.freq_dim, timestep).float(), W, "vggt.time_embedding.0")),
               W, "vggt.time_embedding.2")
    e0 = linear(F.silu(e), W, "vggt.time_projection.1").unflatten(1, (6, cfg.vggt_dim))
    cam, reg = W["vggt.aggregator.camera_token"], W["vggt.aggregator.register_token"]

    def sef(tok):   # slice_expand_and_flatten, aggregator.py:283-306 (B = 1)
        return torch.cat([tok[:, 0:1], tok[:, 1:].expand(1, f - 1, *tok.shape[2:])], dim=1)[0]

    cam_tok = sef(cam)
    if camera_token is not None:
        # CamTokenProjector.forward (vggt/layers/block.py:286-297; aggregator.py:265-266): pad with 3 copies of the first pose,
        # group 4 poses x 9 numbers per latent frame, MLP 36 -> 128 -> GELU(erf) -> C: one camera token per frame
        pre = "vggt.aggregator.CamTokenProjector.mlp."
        c = torch.cat([camera_token, camera_token[:, :1].repeat(1, 3, 1)], dim=1)
        c = c.view(1, c.shape[1] // 4, 36).flatten(0, 1)
        cam_tok = linear(F.gelu(linear(c, W, pre + "0")), W, pre + "2").view(-1, 1, cfg.vggt_dim)
    tokens = torch.cat([cam_tok, sef(reg), pt], dim=1)                       # [f, P, C]
    P = tokens.shape[1]
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    pos = torch.stack([ys.reshape(-1), xs.reshape(-1)], dim=-1) + 1
    pos = torch.cat([torch.zeros(cfg.n_special, 2, dtype=pos.dtype), pos], dim=0)
    pos = pos.unsqueeze(0).expand(f, -1, -1)
    if collect is not None:
        collect["tokens_in"] = tokens.reshape(f * P, -1).clone()

    for i in range(cfg.n_irg):
        fp = f"vggt.aggregator.frame_blocks.{i}."
        tokens, e_f = vggt_block_partial(tokens, pos, e0, W, fp, cfg)         # [f, P, C], frame attention
        tokens = vggt_block_remaining(tokens, e_f, W, fp, cfg)
        frame_out = tokens
        p = cfg.dit_prefix(cfg.start_index + i)
        gp = cfg.global_prefix(i)
        adapter = cfg.has_adapter(cfg.start_index + i)
        tg = tokens.reshape(1, f * P, -1)
        pg = pos.reshape(1, f * P, 2)
        x, mods = dit_block_partial(x, ctx, t_mod, freqs, W, p, cfg, adapter, plucker_fea)
        tg, e_g = vggt_block_partial(tg, pg, e0, W, gp, cfg)
        if i in cfg.cross_attention_list and not uncond:
            x, tg = bicross(x, tg, freqs_bi_dit, freqs_bi_agg, W, f"IRGBlock.{i}.bicross_attention.", cfg)
        x = dit_block_remaining(x, mods, W, p, cfg)
        tg = vggt_block_remaining(tg, e_g, W, gp, cfg)
        tokens = tg.reshape(f, P, -1)
        if per_block is not None:
            per_block("x", cfg.start_index + i, x[0])
            per_block("tok", i, tokens.reshape(f * P, -1))
        if collect is not None and "output_list" in collect:
            # model_wan21.py:208-212: frame | global intermediates of every layer, concatenated on channels -> [f, P, 2C]
            collect["output_list"][i] = torch.cat([frame_out, tokens], dim=-1)
    if collect is not None:
        collect["x_final"] = x[0].clone()
        collect["tokens_final"] = tokens.reshape(f * P, -1).clone()

    # head (wan_video_dit.py:344-358) + unpatchify (:437-442)
    sh, sc = (W[pd + "head.modulation"] + t).chunk(2, dim=1)
    x = linear(F.layer_norm(x, (cfg.dim,), None, None, cfg.eps) * (1 + sc) + sh, W, pd + "head.head")
    x = x.view(1, f, h, w, 1, 2, 2, cfg.out_dim).permute(0, 7, 1, 4, 2, 5, 3, 6).reshape(1, cfg.out_dim, f, 2 * h, 2 * w)
    return x
