"""TEST INFRASTRUCTURE ONLY -- CPU restatement (PyTorch fp32) of the Wan video VAE decoder.

`WanVideoVAE.decode` (FantasyWorld/diffsynth_wan21/models/wan_video_vae.py:776-782) turns the final latents into frames once per
generation; its tiling / blending (`tiled_decode`, :643-692) calls `VideoVAE_.decode(z, scale)` (:552-575) per tile, which is
what is restated here: un-normalise, conv2, then Decoder3d (:379-482) frame by frame through a convolution cache.

As for the geometry heads' up-sampler (oracle/fw_heads_oracle.py) the cached, chunked evaluation equals whole-sequence causal
convolutions: every CausalConv3d sees the two previous frames of its own input (zeros before the start); the temporal
up-sampler passes frame 0 through and convolves frames 1.. with zero history (:120-156).  The pin test checks that against the
chunked reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import torch
import torch.nn.functional as F

from .fw_heads_oracle import causal_conv3d, chan_rms_norm

VAE_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497,
            0.2503, -0.2921]
VAE_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251,
           1.9160]


def residual_block(x, W, pre):
    # wan_video_vae.py:198-232; x [C, T, H, W]
    h = x if (pre + "shortcut.weight") not in W else causal_conv3d(x, W[pre + "shortcut.weight"], W[pre + "shortcut.bias"])
    y = F.silu(chan_rms_norm(x, W[pre + "residual.0.gamma"]))
    y = causal_conv3d(y, W[pre + "residual.2.weight"], W[pre + "residual.2.bias"])
    y = F.silu(chan_rms_norm(y, W[pre + "residual.3.gamma"]))
    y = causal_conv3d(y, W[pre + "residual.6.weight"], W[pre + "residual.6.bias"])
    return y + h


def attention_block(x, W, pre):
    # wan_video_vae.py:235-273: per frame, one head of width C over the h*w positions
    C, T, H, Wd = x.shape
    y = chan_rms_norm(x, W[pre + "norm.gamma"]).permute(1, 0, 2, 3)                     # [T, C, H, W]
    qkv = F.conv2d(y, W[pre + "to_qkv.weight"], W[pre + "to_qkv.bias"]).reshape(T, 3 * C, H * Wd).permute(0, 2, 1)
    q, k, v = qkv.chunk(3, dim=-1)
    a = torch.softmax(q @ k.transpose(1, 2) / C ** 0.5, dim=-1) @ v                    # [T, hw, C]
    a = a.permute(0, 2, 1).reshape(T, C, H, Wd)
    a = F.conv2d(a, W[pre + "proj.weight"], W[pre + "proj.bias"])
    return x + a.permute(1, 0, 2, 3)


def resample(x, W, pre, temporal):
    # wan_video_vae.py:82-156, modes 'upsample3d' / 'upsample2d'
    C, T = x.shape[:2]
    if temporal and T > 1:
        y = causal_conv3d(x[:, 1:], W[pre + "time_conv.weight"], W[pre + "time_conv.bias"])
        y = torch.stack((y[:C], y[C:]), dim=2).reshape(C, 2 * (T - 1), *x.shape[2:])
        x = torch.cat([x[:, :1], y], dim=1)
    y = F.interpolate(x.permute(1, 0, 2, 3), scale_factor=(2.0, 2.0), mode="nearest-exact")
    y = F.conv2d(y, W[pre + "resample.1.weight"], W[pre + "resample.1.bias"], padding=1)
    return y.permute(1, 0, 2, 3)


def vae_decode(W, z, pre="", dim_mult=(1, 2, 4, 4), num_res_blocks=2, temporal_upsample=(True, True, False)):
    """z [1, 16, T, h, w] (normalised latents) -> video [1, 3, 4T-3, 8h, 8w]: VideoVAE_.decode(z, [mean, 1/std]) (:552-575)."""
    mean, std = torch.tensor(VAE_MEAN).view(-1, 1, 1, 1), torch.tensor(VAE_STD).view(-1, 1, 1, 1)
    x = z[0] / (1.0 / std) + mean
    x = causal_conv3d(x, W[pre + "conv2.weight"], W[pre + "conv2.bias"])
    d = pre + "decoder."
    x = causal_conv3d(x, W[d + "conv1.weight"], W[d + "conv1.bias"])
    x = residual_block(x, W, d + "middle.0.")
    x = attention_block(x, W, d + "middle.1.")
    x = residual_block(x, W, d + "middle.2.")
    idx = 0
    for i in range(len(dim_mult)):
        for _ in range(num_res_blocks + 1):
            x = residual_block(x, W, f"{d}upsamples.{idx}.")
            idx += 1
        if i != len(dim_mult) - 1:
            x = resample(x, W, f"{d}upsamples.{idx}.", temporal_upsample[i])
            idx += 1
    x = F.silu(chan_rms_norm(x, W[d + "head.0.gamma"]))
    x = causal_conv3d(x, W[d + "head.2.weight"], W[d + "head.2.bias"])
    return x[None]
