"""Deterministic synthetic weights / inputs for the fusion model ("random weights" of BASELINE.json).

Parameter names are the reference's own (relative to FantasyWorldFusionModel), so the same dictionary loads into
the reference module tree (oracle/ref_harness.py, build container only), into the CPU oracle and into the HIP engine.
Every parameter gets its own generator seeded from crc32(name): the values do not depend on enumeration order,
device or the set of other parameters.  Zero-initialised reference parameters (gamma_m1/m2, adapter last layer,
FantasyWorld/fusion/layer/block.py:173-174, camera_control.py:53-56) are randomised too, otherwise the cross paths
would be invisible to a parity test (SURVEY.md 8 trap 11).
"""
import math
import zlib
from collections import OrderedDict

import torch

from .config import FWConfig, HeadsConfig


def _lin(spec, name, n_out, n_in, gain=1.0):
    spec[name + ".weight"] = ((n_out, n_in), ("normal", gain / math.sqrt(n_in)))
    spec[name + ".bias"] = ((n_out,), ("normal", 0.02))


def _dit_block(spec, cfg: FWConfig, pre: str, b: int):
    d = cfg.dim
    spec[pre + "modulation"] = ((1, 6, d), ("normal", 1.0 / math.sqrt(d)))
    for n in ("q", "k", "v", "o"):
        _lin(spec, pre + "self_attn." + n, d, d)
    spec[pre + "self_attn.norm_q.weight"] = ((d,), ("ones_normal", 0.1))
    spec[pre + "self_attn.norm_k.weight"] = ((d,), ("ones_normal", 0.1))
    for n in ("q", "k", "v", "o"):
        _lin(spec, pre + "cross_attn." + n, d, d)
    spec[pre + "cross_attn.norm_q.weight"] = ((d,), ("ones_normal", 0.1))
    spec[pre + "cross_attn.norm_k.weight"] = ((d,), ("ones_normal", 0.1))
    if cfg.has_image_input:
        _lin(spec, pre + "cross_attn.k_img", d, d)
        _lin(spec, pre + "cross_attn.v_img", d, d)
        spec[pre + "cross_attn.norm_k_img.weight"] = ((d,), ("ones_normal", 0.1))
    if cfg.has_adapter(b):
        p = pre + "cross_attn.processor."
        _lin(spec, p + "k_proj.group1", cfg.plucker_dim, cfg.plucker_dim)
        _lin(spec, p + "k_proj.group2.0", cfg.adapter_hidden, d)
        _lin(spec, p + "k_proj.group2.2", cfg.plucker_dim, cfg.adapter_hidden)
        _lin(spec, p + "v_proj.group2.0", cfg.adapter_reduced, cfg.plucker_dim)
        _lin(spec, p + "v_proj.group2.2", d, cfg.adapter_reduced, gain=0.5)
    spec[pre + "norm3.weight"] = ((d,), ("ones_normal", 0.1))
    spec[pre + "norm3.bias"] = ((d,), ("normal", 0.05))
    _lin(spec, pre + "ffn.0", cfg.ffn_dim, d)
    _lin(spec, pre + "ffn.2", d, cfg.ffn_dim)


def _vggt_block(spec, cfg: FWConfig, pre: str):
    c = cfg.vggt_dim
    hd = c // cfg.vggt_heads
    spec[pre + "modulation"] = ((1, 6, c), ("normal", 1.0 / math.sqrt(c)))
    spec[pre + "norm1.weight"] = ((c,), ("ones_normal", 0.1))
    spec[pre + "norm1.bias"] = ((c,), ("normal", 0.05))
    _lin(spec, pre + "attn.qkv", 3 * c, c)
    for n in ("q_norm", "k_norm"):
        spec[pre + f"attn.{n}.weight"] = ((hd,), ("ones_normal", 0.1))
        spec[pre + f"attn.{n}.bias"] = ((hd,), ("normal", 0.05))
    _lin(spec, pre + "attn.proj", c, c)
    spec[pre + "ls1.gamma"] = ((c,), ("uniform", 0.3, 0.8))
    spec[pre + "norm2.weight"] = ((c,), ("ones_normal", 0.1))
    spec[pre + "norm2.bias"] = ((c,), ("normal", 0.05))
    _lin(spec, pre + "mlp.fc1", cfg.vggt_mlp, c)
    _lin(spec, pre + "mlp.fc2", c, cfg.vggt_mlp)
    spec[pre + "ls2.gamma"] = ((c,), ("uniform", 0.3, 0.8))


def weight_spec(cfg: FWConfig) -> "OrderedDict[str, tuple]":
    """name -> (shape, init) for every parameter on the per-step hot path."""
    assert cfg.cross_attention_list == list(range(len(cfg.cross_attention_list))), \
        "only prefix-contiguous cross_attention_list is meaningful in the reference (model_wan21.py:188-190)"
    spec = OrderedDict()
    d = cfg.dim
    pd = "pipe.dit."
    spec[pd + "patch_embedding.weight"] = ((d, cfg.in_dim, 1, 2, 2), ("normal", 1.0 / math.sqrt(cfg.in_dim * 4)))
    spec[pd + "patch_embedding.bias"] = ((d,), ("normal", 0.02))
    _lin(spec, pd + "text_embedding.0", d, cfg.text_dim)
    _lin(spec, pd + "text_embedding.2", d, d)
    _lin(spec, pd + "time_embedding.0", d, cfg.freq_dim)
    _lin(spec, pd + "time_embedding.2", d, d)
    _lin(spec, pd + "time_projection.1", 6 * d, d)
    for b in range(cfg.num_layers):
        _dit_block(spec, cfg, cfg.dit_prefix(b), b)
    spec[pd + "head.modulation"] = ((1, 2, d), ("normal", 1.0 / math.sqrt(d)))
    _lin(spec, pd + "head.head", cfg.out_dim * 4, d)
    if cfg.has_image_input:
        spec[pd + "img_emb.proj.0.weight"] = ((cfg.clip_dim,), ("ones_normal", 0.1))
        spec[pd + "img_emb.proj.0.bias"] = ((cfg.clip_dim,), ("normal", 0.05))
        _lin(spec, pd + "img_emb.proj.1", cfg.clip_dim, cfg.clip_dim)
        _lin(spec, pd + "img_emb.proj.3", d, cfg.clip_dim)
        spec[pd + "img_emb.proj.4.weight"] = ((d,), ("ones_normal", 0.1))
        spec[pd + "img_emb.proj.4.bias"] = ((d,), ("normal", 0.05))
    if cfg.control_adapter:
        ca = pd + "control_adapter."
        kin = cfg.control_in_dim * 64
        spec[ca + "conv.weight"] = ((d, kin, 2, 2), ("normal", 1.0 / math.sqrt(kin * 4)))
        spec[ca + "conv.bias"] = ((d,), ("normal", 0.02))
        for n in ("conv1", "conv2"):
            spec[ca + f"residual_blocks.0.{n}.weight"] = ((d, d, 3, 3), ("normal", 1.0 / math.sqrt(d * 9)))
            spec[ca + f"residual_blocks.0.{n}.bias"] = ((d,), ("normal", 0.02))
    c = cfg.vggt_dim
    spec["vggt.projection_head.weight"] = ((c, d, 1, 1, 1), ("normal", 1.0 / math.sqrt(d)))
    spec["vggt.projection_head.bias"] = ((c,), ("normal", 0.02))
    spec["vggt.aggregator.camera_token"] = ((1, 2, 1, c), ("normal", 0.5))
    spec["vggt.aggregator.register_token"] = ((1, 2, cfg.n_special - 1, c), ("normal", 0.5))
    # CamTokenProjector (vggt/layers/block.py:276-297): 4 poses x 9 -> 128 -> C, only used when joint_forward gets camera_token
    _lin(spec, "vggt.aggregator.CamTokenProjector.mlp.0", 128, 36)
    _lin(spec, "vggt.aggregator.CamTokenProjector.mlp.2", c, 128)
    _lin(spec, "vggt.time_embedding.0", c, cfg.freq_dim)
    _lin(spec, "vggt.time_embedding.2", c, c)
    _lin(spec, "vggt.time_projection.1", 6 * c, c)
    for j in range(cfg.n_irg):
        _vggt_block(spec, cfg, f"vggt.aggregator.frame_blocks.{j}.")
        _vggt_block(spec, cfg, cfg.global_prefix(j))
    for j in range(len(cfg.cross_attention_list)):
        p = f"IRGBlock.{j}.bicross_attention."
        spec[p + "gamma_m1"] = ((d,), ("uniform", 0.3, 0.8))
        spec[p + "gamma_m2"] = ((c,), ("uniform", 0.3, 0.8))
        bd = cfg.bicross_dim
        _lin(spec, p + "cross_attn.m1_proj", bd, d)
        _lin(spec, p + "cross_attn.m2_proj", bd, c)
        _lin(spec, p + "cross_attn.values_m1_proj", bd, d)
        _lin(spec, p + "cross_attn.values_m2_proj", bd, c)
        _lin(spec, p + "cross_attn.out_m1_proj", d, bd)
        _lin(spec, p + "cross_attn.out_m2_proj", c, bd)
    return spec


def make_param(name, shape, init, device="cpu", dtype=torch.float32, seed=0):
    g = torch.Generator(device=device)
    g.manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
    kind = init[0]
    if kind == "normal":
        t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * init[1]
    elif kind == "ones_normal":
        t = 1.0 + torch.randn(shape, generator=g, device=device, dtype=torch.float32) * init[1]
    elif kind == "uniform":
        t = init[1] + (init[2] - init[1]) * torch.rand(shape, generator=g, device=device, dtype=torch.float32)
    else:
        raise ValueError(kind)
    return t.to(dtype)


def make_weights(cfg: FWConfig, device="cpu", dtype=torch.float32, seed=0, bf16_round=True):
    """All hot-path parameters.  With device='cpu' the values are bit-reproducible across machines (same torch).

    bf16_round=True rounds every tensor to bf16-representable values (kept in `dtype`): the reference loads its
    checkpoint in bf16 (model_wan21.py:38-41,101), and the oracle, the golden fixtures and the HIP engine must all
    see the same parameter values.
    """
    out = OrderedDict()
    for name, (shape, init) in weight_spec(cfg).items():
        t = make_param(name, shape, init, device=device, dtype=torch.float32, seed=seed)
        if bf16_round:
            t = t.to(torch.bfloat16).to(torch.float32)
        out[name] = t.to(dtype)
    return out


class LazyWeights:
    """name -> tensor mapping that draws each parameter on demand, ON `device` (same per-name generators as make_weights: the values
    depend on (name, seed) only).  For the full-depth models -- 16 B parameters = 64 GB in fp32 -- where a materialised dictionary
    next to the module tree that clones from it would double the footprint: bench.py packs straight from it, the full-depth parity
    test (tests/test_full_depth_gpu.py) loads the reference module tree from it.  `specs`: one or more name -> (shape, init) tables
    (weight_spec, heads_weight_spec, pose_encoder_weight_spec)."""

    def __init__(self, *specs, device="cpu", seed=0, bf16_round=True):
        self.spec = OrderedDict()
        for s in specs:
            self.spec.update(s)
        self.device, self.seed, self.bf16_round = device, seed, bf16_round

    def __contains__(self, name):
        return name in self.spec

    def __iter__(self):
        return iter(self.spec)

    def __len__(self):
        return len(self.spec)

    def keys(self):
        return self.spec.keys()

    def __getitem__(self, name):
        shape, init = self.spec[name]
        t = make_param(name, shape, init, device=self.device, dtype=torch.float32, seed=self.seed)
        return t.to(torch.bfloat16).to(torch.float32) if self.bf16_round else t

    def items(self):
        return ((k, self[k]) for k in self.spec)


def make_inputs(cfg: FWConfig, f: int, h2: int, w2: int, seed=1, device="cpu", dtype=torch.float32,
                text_len=512, timestep=500.0):
    """Synthetic joint_forward inputs: latents [1,16,f,h2,w2], y [1,20,f,h2,w2] (4 mask + 16 latent
    channels, first latent frame masked in as wan_video.py:237-262 does), context [1,512,4096] (pos and neg draw),
    clip_feature [1,257,1280], plucker_fea [1,L,2048], plucker_context_lens [f]."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float32)
    L = f * (h2 // 2) * (w2 // 2)
    x = r(1, cfg.out_dim, f, h2, w2)
    mask = torch.zeros(1, 4, f, h2, w2)
    mask[:, :, 0] = 1.0
    y = torch.cat([mask, r(1, cfg.in_dim - cfg.out_dim - 4, f, h2, w2)], dim=1)
    ins = dict(
        x=x, y=y,
        context=r(1, text_len, cfg.text_dim),
        context_neg=r(1, text_len, cfg.text_dim),
        clip_feature=r(1, cfg.clip_tokens, cfg.clip_dim) if cfg.has_image_input else None,
        plucker_fea=r(1, L, cfg.plucker_dim) if cfg.camera_adapter else None,
        # Wan2.2: Pluecker map folded to 24 channels at pixel resolution (inference_wan22.py:204-218): [1, 24, f, 8*h2, 8*w2]
        control_camera_latents_input=r(1, cfg.control_in_dim, f, 8 * h2, 8 * w2) if cfg.control_adapter else None,
        timestep=torch.tensor([timestep], dtype=torch.float32),
        # optional camera_token [1, 4 (f - 1) + 1, 9] (one pose encoding per video frame; CamTokenProjector groups 4 per latent frame)
        camera_token=r(1, 4 * (f - 1) + 1, 9),
    )
    lens = torch.ones(f, dtype=torch.long)
    lens[1:] = 4
    ins["plucker_context_lens"] = lens
    out = {}
    for k, v in ins.items():
        if v is None or v.dtype == torch.long:
            out[k] = v if v is None else v.to(device)
        else:
            # round to bf16-representable values: what the reference's bf16 inference path feeds joint_forward
            out[k] = v.to(torch.bfloat16).to(dtype).to(device)
    return out


# ---------------------------------------------------------------------------------------------------------------
# VGGT geometry heads: vggt/heads/camera_head.py, vggt/heads/dpt_head.py, wan/modules/vae_modified.py
# ---------------------------------------------------------------------------------------------------------------
def _conv(spec, name, shape, bias=True, gain=1.0):
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    spec[name + ".weight"] = (tuple(shape), ("normal", gain / math.sqrt(fan_in)))
    if bias:
        spec[name + ".bias"] = ((shape[0],), ("normal", 0.02))


def heads_weight_spec(hc: HeadsConfig) -> "OrderedDict[str, tuple]":
    """name -> (shape, init) for vggt.camera_head / vggt.depth_head / vggt.point_head (reference parameter names).
    LayerScale gammas, the empty pose token and the pose-embedding are drawn at working scale (the reference initialises
    them at 0.01 / zeros, which would hide the trunk from a parity test)."""
    spec = OrderedDict()
    C = hc.dim_in
    p = "vggt.camera_head."
    for b in range(hc.trunk_depth):
        q = f"{p}trunk.{b}."
        spec[q + "modulation"] = ((1, 6, C), ("normal", 1.0 / math.sqrt(C)))          # unused by the head (block.py:60)
        spec[q + "norm1.weight"] = ((C,), ("ones_normal", 0.1))
        spec[q + "norm1.bias"] = ((C,), ("normal", 0.05))
        _lin(spec, q + "attn.qkv", 3 * C, C)
        _lin(spec, q + "attn.proj", C, C)
        spec[q + "ls1.gamma"] = ((C,), ("normal", 0.3))
        spec[q + "norm2.weight"] = ((C,), ("ones_normal", 0.1))
        spec[q + "norm2.bias"] = ((C,), ("normal", 0.05))
        _lin(spec, q + "mlp.fc1", C * hc.cam_mlp_ratio, C)
        _lin(spec, q + "mlp.fc2", C, C * hc.cam_mlp_ratio)
        spec[q + "ls2.gamma"] = ((C,), ("normal", 0.3))
    for n in ("token_norm", "trunk_norm"):
        spec[p + n + ".weight"] = ((C,), ("ones_normal", 0.1))
        spec[p + n + ".bias"] = ((C,), ("normal", 0.05))
    spec[p + "empty_pose_tokens"] = ((1, 1, 9), ("normal", 0.5))
    _lin(spec, p + "embed_pose", C, 9)
    _lin(spec, p + "poseLN_modulation.1", 3 * C, C, gain=0.5)
    _conv(spec, p + "camera_time_upsample.expand_channels", (4 * C, C, 1))
    _lin(spec, p + "pose_branch.fc1", C // 2, C)
    _lin(spec, p + "pose_branch.fc2", 9, C // 2)
    for head, odim in (("vggt.depth_head.", hc.depth_out), ("vggt.point_head.", hc.point_out)):
        spec[head + "norm.weight"] = ((C,), ("ones_normal", 0.1))
        spec[head + "norm.bias"] = ((C,), ("normal", 0.05))
        oc = hc.out_channels
        for i, c in enumerate(oc):
            _conv(spec, head + f"projects.{i}", (c, C, 1, 1))
        _conv(spec, head + "resize_layers.0", (oc[0], oc[0], 4, 4), gain=4.0)     # ConvTranspose2d: [in, out, k, k], k*k taps disjoint
        _conv(spec, head + "resize_layers.1", (oc[1], oc[1], 2, 2), gain=2.0)
        _conv(spec, head + "resize_layers.3", (oc[3], oc[3], 3, 3))
        for i, c in enumerate(oc):
            t = head + f"temporal_upsamplers.{i}."
            _conv(spec, t + "conv2", (c, c, 1, 1, 1))
            for u in (0, 2):
                _conv(spec, t + f"decoder.upsamples.{u}.time_conv", (2 * c, c, 3, 1, 1))
            for u in (1, 3):
                spec[t + f"decoder.upsamples.{u}.residual.0.gamma"] = ((c, 1, 1, 1), ("ones_normal", 0.1))
                _conv(spec, t + f"decoder.upsamples.{u}.residual.2", (c, c, 3, 3, 3))
        sc = head + "scratch."
        f = hc.features
        for i, c in enumerate(oc):
            _conv(spec, sc + f"layer{i + 1}_rn", (f, c, 3, 3), bias=False)
        for r in (1, 2, 3, 4):
            q = sc + f"refinenet{r}."
            _conv(spec, q + "out_conv", (f, f, 1, 1))
            for u in ((1, 2) if r != 4 else (2,)):
                _conv(spec, q + f"resConfUnit{u}.conv1", (f, f, 3, 3))
                _conv(spec, q + f"resConfUnit{u}.conv2", (f, f, 3, 3))
        _conv(spec, sc + "output_conv1", (f // 2, f, 3, 3))
        _conv(spec, sc + "output_conv2.0", (32, f // 2, 3, 3))
        _conv(spec, sc + "output_conv2.2", (odim, 32, 1, 1), gain=0.5)
    return spec


def make_heads_weights(hc: HeadsConfig, device="cpu", dtype=torch.float32, seed=0, bf16_round=True):
    out = OrderedDict()
    for name, (shape, init) in heads_weight_spec(hc).items():
        t = make_param(name, shape, init, device=device, dtype=torch.float32, seed=seed)
        if bf16_round:
            t = t.to(torch.bfloat16).to(torch.float32)
        out[name] = t.to(dtype)
    return out


def make_output_list(hc: HeadsConfig, S: int, ph: int, pw: int, n_special=5, seed=3, device="cpu"):
    """Synthetic aggregator output_list: layer -> fp32 [S, n_special + ph*pw, dim_in], unit-scale tokens (the residual
    streams of the aggregator are O(1)-O(10); the heads normalise them first)."""
    need = sorted(set(hc.layer_idx) | {max(hc.layer_idx)})
    out = {}
    for layer in need:
        g = torch.Generator(device="cpu").manual_seed(1000 * seed + layer)
        out[layer] = (torch.randn(S, n_special + ph * pw, hc.dim_in, generator=g) * 2.0).to(device)
    return out


def pose_encoder_weight_spec(dim=5120, context_dim=2048, in_channels=6, pre="camera_condition.pose_encoder."):
    """CameraPoseEncoder parameters (diffsynth_wan21/models/pose_adaptor_ac3d.py:10-48)."""
    spec = OrderedDict()
    c0 = in_channels * 64
    p = pre + "controlnet_encode_first."
    _conv(spec, p + "0", (c0, c0, 1, 1))
    _conv(spec, p + "2", (c0, c0, 1, 1))
    for n, c in (("1", c0), ("3", c0)):
        spec[p + n + ".weight"] = ((c,), ("ones_normal", 0.1))
        spec[p + n + ".bias"] = ((c,), ("normal", 0.05))
    p = pre + "controlnet_encode_second."
    _conv(spec, p + "0", (2 * c0, c0, 1, 1))
    spec[p + "1.weight"] = ((2 * c0,), ("ones_normal", 0.1))
    spec[p + "1.bias"] = ((2 * c0,), ("normal", 0.05))
    _conv(spec, pre + "patch_embedding", (dim, 2 * c0, 1, 2, 2))
    p = pre + "fc."
    _lin(spec, p + "0", dim // 2, dim)
    _lin(spec, p + "3", context_dim, dim // 2)
    for n, c in (("1", dim // 2), ("4", context_dim)):
        spec[p + n + ".weight"] = ((c,), ("ones_normal", 0.1))
        spec[p + n + ".bias"] = ((c,), ("normal", 0.05))
    return spec


def make_pose_encoder_weights(device="cpu", seed=0, bf16_round=True, **kw):
    out = OrderedDict()
    for name, (shape, init) in pose_encoder_weight_spec(**kw).items():
        t = make_param(name, shape, init, device=device, dtype=torch.float32, seed=seed)
        out[name] = t.to(torch.bfloat16).to(torch.float32) if bf16_round else t
    return out


def make_plucker(frames, H, W, channels=6, seed=5, device="cpu"):
    """Synthetic Pluecker embedding [1, frames, H, W, 6] (unit-scale ray moments / directions), bf16-representable."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(1, frames, H, W, channels, generator=g).to(torch.bfloat16).float().to(device)


def vae_decoder_weight_spec(dim=96, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2, temporal_upsample=(True, True, False),
                            pre=""):
    """conv2 + Decoder3d parameters of VideoVAE_ (diffsynth_wan21/models/wan_video_vae.py:379-430, 492-517), names relative
    to the VideoVAE_ module (`pipe.vae.model.` on the fusion model)."""
    spec = OrderedDict()
    dims = [dim * u for u in [dim_mult[-1]] + list(dim_mult[::-1])]
    _conv(spec, pre + "conv2", (z_dim, z_dim, 1, 1, 1))
    d = pre + "decoder."
    _conv(spec, d + "conv1", (dims[0], z_dim, 3, 3, 3))

    def res(p, cin, cout):
        spec[p + "residual.0.gamma"] = ((cin, 1, 1, 1), ("ones_normal", 0.1))
        _conv(spec, p + "residual.2", (cout, cin, 3, 3, 3))
        spec[p + "residual.3.gamma"] = ((cout, 1, 1, 1), ("ones_normal", 0.1))
        _conv(spec, p + "residual.6", (cout, cout, 3, 3, 3))
        if cin != cout:
            _conv(spec, p + "shortcut", (cout, cin, 1, 1, 1))

    res(d + "middle.0.", dims[0], dims[0])
    spec[d + "middle.1.norm.gamma"] = ((dims[0], 1, 1), ("ones_normal", 0.1))
    _conv(spec, d + "middle.1.to_qkv", (3 * dims[0], dims[0], 1, 1))
    _conv(spec, d + "middle.1.proj", (dims[0], dims[0], 1, 1))              # zero-initialised in the reference: randomised here
    res(d + "middle.2.", dims[0], dims[0])
    idx = 0
    out_dim = dims[0]
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        if i in (1, 2, 3):
            cin = cin // 2
        for _ in range(num_res_blocks + 1):
            res(f"{d}upsamples.{idx}.", cin, cout)
            cin = cout
            idx += 1
        if i != len(dim_mult) - 1:
            p = f"{d}upsamples.{idx}."
            _conv(spec, p + "resample.1", (cout // 2, cout, 3, 3))
            if temporal_upsample[i]:
                _conv(spec, p + "time_conv", (2 * cout, cout, 3, 1, 1))
            idx += 1
        out_dim = cout
    spec[d + "head.0.gamma"] = ((out_dim, 1, 1, 1), ("ones_normal", 0.1))
    _conv(spec, d + "head.2", (3, out_dim, 3, 3, 3))
    return spec


def make_vae_decoder_weights(device="cpu", seed=0, bf16_round=True, **kw):
    out = OrderedDict()
    for name, (shape, init) in vae_decoder_weight_spec(**kw).items():
        t = make_param(name, shape, init, device=device, dtype=torch.float32, seed=seed)
        out[name] = t.to(torch.bfloat16).to(torch.float32) if bf16_round else t
    return out


def make_latents(T, h, w, z_dim=16, seed=7, device="cpu"):
    """Synthetic normalised latents [1, 16, T, h, w] (what the sampler hands to the VAE), bf16-representable."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(1, z_dim, T, h, w, generator=g).to(torch.bfloat16).float().to(device)
