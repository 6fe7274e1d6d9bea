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

from .config import FWConfig


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
