"""TEST INFRASTRUCTURE ONLY -- CPU restatement (PyTorch fp32) of FantasyWorld's VGGT geometry heads.

`VGGT._head_predction` (FantasyWorld/vggt/models/vggt.py:134-154) runs once per generation, on the `output_list` of the
last denoising step: CameraHead -> pose_enc, DPTHead_3D_Causal x2 -> depth / world points with confidences.  This file is a
functional, module-free restatement; every function cites the reference lines it follows (paths relative to
FantasyWorld/).  It is pinned against the REAL reference modules (tests/test_oracle_pin.py, in the build container) and
against the committed outputs of those modules (tests/golden/heads_*.pt, anywhere).

The reference decodes time frame by frame with a convolution cache (wan/modules/vae_modified.py:454-476).  Reading the
cache logic (vae_modified.py:87-130, 207-226): every CausalConv3d sees exactly the two previous frames of ITS OWN input
sequence, zeros before the start; the temporal up-sampler passes the first frame through and convolves the sequence of
the remaining frames with zero history.  So the chunked decode equals whole-sequence causal convolutions, which is how it
is written here (and the pin test checks that claim against the chunked reference).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import math

import torch
import torch.nn.functional as F


def _lin(x, W, name):
    return F.linear(x, W[name + ".weight"], W.get(name + ".bias"))


# ---------------------------------------------------------------------------------------------------------------
# camera head (vggt/heads/camera_head.py)
# ---------------------------------------------------------------------------------------------------------------
def channel_expand_and_reshape(x, W, pre):
    # wan/modules/vae_modified.py:558-572: Conv1d(C, 4C, 1) over tokens, then a raw reshape [4C, N] -> [C, 4N]
    N, C = x.shape
    y = F.linear(x, W[pre + "expand_channels.weight"].reshape(4 * C, C), W[pre + "expand_channels.bias"])   # [N, 4C]
    return y.t().contiguous().reshape(C, 4 * N).t()                                                        # [4N, C]


def vggt_plain_block(x, W, pre, heads):
    # vggt/layers/block.py:78-120 with e0=None (nn.Sequential call), attention.py:46-69, mlp.py, layer_scale.py
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), W[pre + "norm1.weight"], W[pre + "norm1.bias"], 1e-5)
    qkv = _lin(h, W, pre + "attn.qkv").view(-1, 3, heads, C // heads)
    q, k, v = (qkv[:, i].transpose(0, 1) for i in range(3))                       # [heads, N, hd]
    a = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(C // heads), dim=-1) @ v
    a = _lin(a.transpose(0, 1).reshape(-1, C), W, pre + "attn.proj")
    x = x + a * W[pre + "ls1.gamma"]
    h = F.layer_norm(x, (C,), W[pre + "norm2.weight"], W[pre + "norm2.bias"], 1e-5)
    h = _lin(F.gelu(_lin(h, W, pre + "mlp.fc1")), W, pre + "mlp.fc2")
    return x + h * W[pre + "ls2.gamma"]


def camera_head(W, tokens_last, hc, pre="vggt.camera_head.", num_iterations=4):
    """tokens_last: [S, P, 2C] (last entry of the aggregator's output_list, batch 1) -> list of pose_enc [(S-1)*4+1, 9].

    camera_head.py:76-145.  The time up-sampled tokens are NOT normalised by token_norm (camera_head.py:93, as written)."""
    C = tokens_last.shape[-1]
    pose = tokens_last[:, 0]                                                        # camera token of every latent frame
    up = channel_expand_and_reshape(pose[1:], W, pre + "camera_time_upsample.")
    first = F.layer_norm(pose[:1], (C,), W[pre + "token_norm.weight"], W[pre + "token_norm.bias"], 1e-5)
    tok = torch.cat([first, up], dim=0)                                             # [T, C]
    T = tok.shape[0]
    pred, outs = None, []
    for _ in range(num_iterations):
        if pred is None:
            inp = _lin(W[pre + "empty_pose_tokens"].reshape(1, -1).expand(T, -1), W, pre + "embed_pose")
        else:
            inp = _lin(pred, W, pre + "embed_pose")
        shift, scale, gate = _lin(F.silu(inp), W, pre + "poseLN_modulation.1").chunk(3, dim=-1)
        x = gate * (F.layer_norm(tok, (C,), None, None, 1e-6) * (1 + scale) + shift) + tok
        for b in range(hc.trunk_depth):
            x = vggt_plain_block(x, W, f"{pre}trunk.{b}.", hc.cam_heads)
        x = F.layer_norm(x, (C,), W[pre + "trunk_norm.weight"], W[pre + "trunk_norm.bias"], 1e-5)
        delta = _lin(F.gelu(_lin(x, W, pre + "pose_branch.fc1")), W, pre + "pose_branch.fc2")
        pred = delta if pred is None else pred + delta
        # activate_pose (head_act.py:11-33) with trans "linear", quat "linear", fl "relu" (camera_head.py:36-38)
        outs.append(torch.cat([pred[:, :7], F.relu(pred[:, 7:])], dim=-1))
    return outs


# ---------------------------------------------------------------------------------------------------------------
# DPT head (vggt/heads/dpt_head.py) and its temporal up-sampler (wan/modules/vae_modified.py)
# ---------------------------------------------------------------------------------------------------------------
def uv_pos_embed(C, ph, pw, aspect, ratio=0.1, omega_0=100.0):
    """dpt_head.py:262-283 + heads/utils.py:11-109 -> fp32 [C, ph, pw] (input independent)."""
    diag = (aspect ** 2 + 1.0) ** 0.5
    sx, sy = aspect / diag, 1.0 / diag
    xs = torch.linspace(-sx * (pw - 1) / pw, sx * (pw - 1) / pw, steps=pw, dtype=torch.float32)
    ys = torch.linspace(-sy * (ph - 1) / ph, sy * (ph - 1) / ph, steps=ph, dtype=torch.float32)
    uu, vv = torch.meshgrid(xs, ys, indexing="xy")                                   # [ph, pw]

    def sincos(dim, pos):
        omega = torch.arange(dim // 2, dtype=torch.double) / (dim / 2.0)
        omega = 1.0 / omega_0 ** omega
        out = torch.einsum("m,d->md", pos.reshape(-1), omega)                       # float32 x float64 -> float64
        return torch.cat([torch.sin(out), torch.cos(out)], dim=1).float()

    emb = torch.cat([sincos(C // 2, uu), sincos(C // 2, vv)], dim=-1).view(ph, pw, C)
    return (emb * ratio).permute(2, 0, 1)


def causal_conv3d(x, w, b):
    # vae_modified.py:17-36: zero padding (2*pt in front of time, ph/pw around space); x [C, T, H, W]
    kt, kh, kw = w.shape[2:]
    x = F.pad(x[None], (kw // 2, kw // 2, kh // 2, kh // 2, kt - 1, 0))
    return F.conv3d(x, w, b)[0]


def chan_rms_norm(x, gamma):
    # vae_modified.py:39-54 (images=False): F.normalize over channels * sqrt(C) * gamma; x [C, T, H, W]
    return F.normalize(x, dim=0) * (x.shape[0] ** 0.5) * gamma.reshape(-1, 1, 1, 1)


def upsample3d(x, W, pre):
    # Resample mode 'upsample3d' (vae_modified.py:87-130): first frame unchanged; the others go through
    # CausalConv3d(C, 2C, (3,1,1)) with zero history and are split into two frames each (channels [0:C] first).
    C, T = x.shape[:2]
    if T == 1:
        return x
    y = causal_conv3d(x[:, 1:], W[pre + "time_conv.weight"], W[pre + "time_conv.bias"])          # [2C, T-1, H, W]
    y = torch.stack((y[:C], y[C:]), dim=2).reshape(C, 2 * (T - 1), *x.shape[2:])
    return torch.cat([x[:, :1], y], dim=1)


def residual_block_half(x, W, pre):
    # vae_modified.py:193-226: x + CausalConv3d3x3x3(SiLU(RMS_norm(x))) (in_dim == out_dim -> identity shortcut)
    h = F.silu(chan_rms_norm(x, W[pre + "residual.0.gamma"]))
    return x + causal_conv3d(h, W[pre + "residual.2.weight"], W[pre + "residual.2.bias"])


def temporal_decode(z, W, pre):
    # WanVAE_(location="DPT").decode (vae_modified.py:443-476, Decoder3d_Simple residual=True :391-395); z [C, T, H, W]
    x = causal_conv3d(z, W[pre + "conv2.weight"], W[pre + "conv2.bias"])
    x = upsample3d(x, W, pre + "decoder.upsamples.0.")
    x = residual_block_half(x, W, pre + "decoder.upsamples.1.")
    x = upsample3d(x, W, pre + "decoder.upsamples.2.")
    return residual_block_half(x, W, pre + "decoder.upsamples.3.")


def residual_conv_unit(x, W, pre):
    # dpt_head.py:400-457.  The activation is nn.ReLU(inplace=True) (dpt_head.py:328): it rewrites x, so the skip adds relu(x).
    x = F.relu(x)
    h = F.conv2d(x, W[pre + "conv1.weight"], W[pre + "conv1.bias"], padding=1)
    h = F.conv2d(F.relu(h), W[pre + "conv2.weight"], W[pre + "conv2.bias"], padding=1)
    return h + x


def fusion_block(W, pre, x, skip=None, size=None):
    # FeatureFusionBlock.forward (dpt_head.py:508-536), align_corners=True
    if skip is not None:
        x = x + residual_conv_unit(skip, W, pre + "resConfUnit1.")
    x = residual_conv_unit(x, W, pre + "resConfUnit2.")
    if size is None:
        size = (x.shape[-2] * 2, x.shape[-1] * 2)
    x = F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=True)
    return F.conv2d(x, W[pre + "out_conv.weight"], W[pre + "out_conv.bias"])


def inverse_log_transform(y):
    # head_act.py:115-125
    return torch.sign(y) * torch.expm1(torch.abs(y))


def dpt_head(W, output_list, hc, S, ph, pw, pre, activation, patch_start_idx=5):
    """output_list: layer -> [S, P, 2C] fp32 (batch 1).  Returns (preds [T, H, W, out_dim-1], conf [T, H, W]),
    T = (S-1)*4+1, H = ph*patch, W = pw*patch.  dpt_head.py:133-260 (frame chunking only bounds memory)."""
    H, Wd = ph * hc.dpt_patch, pw * hc.dpt_patch
    aspect = Wd / H
    feats = []
    for i, layer in enumerate(hc.layer_idx):
        x = output_list[layer][:, patch_start_idx:]                                 # [S, ph*pw, 2C]
        C = x.shape[-1]
        x = F.layer_norm(x, (C,), W[pre + "norm.weight"], W[pre + "norm.bias"], 1e-5)
        x = x.permute(0, 2, 1).reshape(S, C, ph, pw)
        x = F.conv2d(x, W[pre + f"projects.{i}.weight"], W[pre + f"projects.{i}.bias"])
        x = x + uv_pos_embed(x.shape[1], ph, pw, aspect)[None]
        rw, rb = W.get(pre + f"resize_layers.{i}.weight"), W.get(pre + f"resize_layers.{i}.bias")
        if i == 0:
            x = F.conv_transpose2d(x, rw, rb, stride=4)
        elif i == 1:
            x = F.conv_transpose2d(x, rw, rb, stride=2)
        elif i == 3:
            x = F.conv2d(x, rw, rb, stride=2, padding=1)
        x = temporal_decode(x.permute(1, 0, 2, 3), W, pre + f"temporal_upsamplers.{i}.")     # [C, T, h, w]
        feats.append(x.permute(1, 0, 2, 3))                                                   # [T, C, h, w]
    sc = pre + "scratch."
    l1, l2, l3, l4 = (F.conv2d(f, W[sc + f"layer{i + 1}_rn.weight"], None, padding=1) for i, f in enumerate(feats))
    out = fusion_block(W, sc + "refinenet4.", l4, None, l3.shape[2:])
    out = fusion_block(W, sc + "refinenet3.", out, l3, l2.shape[2:])
    out = fusion_block(W, sc + "refinenet2.", out, l2, l1.shape[2:])
    out = fusion_block(W, sc + "refinenet1.", out, l1, None)
    out = F.conv2d(out, W[sc + "output_conv1.weight"], W[sc + "output_conv1.bias"], padding=1)
    out = F.interpolate(out, size=(H, Wd), mode="bilinear", align_corners=True)
    out = out + uv_pos_embed(out.shape[1], H, Wd, aspect)[None]
    out = F.relu(F.conv2d(out, W[sc + "output_conv2.0.weight"], W[sc + "output_conv2.0.bias"], padding=1))
    out = F.conv2d(out, W[sc + "output_conv2.2.weight"], W[sc + "output_conv2.2.bias"])
    fmap = out.permute(0, 2, 3, 1)                                                  # activate_head, head_act.py:61-112
    xyz, conf = fmap[..., :-1], fmap[..., -1]
    pts = torch.exp(xyz) if activation == "exp" else inverse_log_transform(xyz)
    return pts, 1 + conf.exp()


def head_prediction(W, output_list, hc, S, ph, pw, patch_start_idx=5):
    """VGGT._head_predction (vggt/models/vggt.py:134-154), batch 1; output_list: layer -> [S, P, 2C]."""
    last = max(output_list.keys())
    pred = {"pose_enc": camera_head(W, output_list[last], hc)[-1][None]}
    d, dc = dpt_head(W, output_list, hc, S, ph, pw, "vggt.depth_head.", "exp", patch_start_idx)
    p, pc = dpt_head(W, output_list, hc, S, ph, pw, "vggt.point_head.", "inv_log", patch_start_idx)
    pred.update(depth=d[None], depth_conf=dc[None], world_points=p[None], world_points_conf=pc[None])
    return pred
