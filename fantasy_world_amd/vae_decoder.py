"""Wan video VAE decoder on the engine's op set: latents -> frames, once per generation.

Replaces `VideoVAE_.decode(z, scale)` (FantasyWorld/diffsynth_wan21/models/wan_video_vae.py:552-575 with Decoder3d :379-482,
ResidualBlock :198-232, AttentionBlock :235-273, Resample :82-156), the call `WanVideoVAE.tiled_decode` / `single_decode`
(:643-692, :752-755) make per tile; the tiling and blending stay the reference's own code.

Same scheme as the geometry heads (fantasy_world_amd.heads): channels-last feature maps [frames*H*W, C] in bf16 with C padded to
a multiple of 64 (16 -> 64, 96 -> 128), every convolution = `fw_im2col` gather + `fw_gemm_bf16` with bias / residual in the
epilogue, whole-sequence causal convolutions instead of the reference's frame-by-frame cache (same numbers: see
oracle/fw_vae_oracle.py).  The x2 nearest-neighbour up-sampling of a Resample block is folded into the gather of the 3x3
convolution that follows it, so the 4x larger map is never written.  The mid-block attention has ONE head of width 384
(outside the flash kernel's 64 / 96 / 128): per frame it is two GEMMs around a row softmax.
"""
import torch

from .convnet import ConvNetBase, _cpad
from .hip_ops import Linear

VAE_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497,
            0.2503, -0.2921]                      # wan_video_vae.py:604-611
VAE_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251,
           1.9160]


class VaeDecoder(ConvNetBase):
    def __init__(self, get, ops, pre="", dim=96, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2,
                 temporal_upsample=(True, True, False), max_col_bytes=4 << 30):
        """get(name) -> parameter of the reference's VideoVAE_ (names relative to it: "conv2.weight", "decoder.conv1.weight" ...)."""
        self.ops, self.max_col_bytes = ops, max_col_bytes
        self.z_dim = z_dim
        g = lambda n: get(pre + n)
        sub = lambda p: (lambda n: get(pre + p + n))
        dims = [dim * u for u in [dim_mult[-1]] + list(dim_mult[::-1])]
        # conv2 (1x1x1) with the latent un-normalisation z * std + mean folded in: W (D z + m) + b = (W D) z + (W m + b)
        w2 = g("conv2.weight").float().reshape(z_dim, z_dim).cpu()
        mean, std = torch.tensor(VAE_MEAN[:z_dim]), torch.tensor(VAE_STD[:z_dim])
        self.default_scale = (mean, 1.0 / std)
        self._w2, self._b2 = w2, g("conv2.bias").float().cpu()
        self._conv2_cache = {}
        self._conv2_ident = None
        d = "decoder."
        self.conv1 = self._conv(sub(d), "conv1")
        self.mid = [self._res(sub(d + "middle.0."), dims[0], dims[0]), self._attn(sub(d + "middle.1."), dims[0]),
                    self._res(sub(d + "middle.2."), dims[0], dims[0])]
        self.stages = []
        idx = 0
        cout = dims[0]
        for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
            if i in (1, 2, 3):
                cin = cin // 2
            blocks = []
            for _ in range(num_res_blocks + 1):
                blocks.append(self._res(sub(f"{d}upsamples.{idx}."), cin, cout))
                cin = cout
                idx += 1
            up = None
            if i != len(dim_mult) - 1:
                s_ = sub(f"{d}upsamples.{idx}.")
                up = dict(conv=self._conv(s_, "resample.1"), time=self._conv(s_, "time_conv") if temporal_upsample[i] else None,
                          c=cout)
                assert cout % 64 == 0, "the temporal up-sampler splits 2C channels in two halves"
                idx += 1
            self.stages.append((blocks, up))
        self.head_gamma = self._vec(g(d + "head.0.gamma"), _cpad(cout))
        self.head_c = cout
        self.head = self._conv(sub(d), "head.2", pad_out=False)

    def _res(self, get, cin, cout):
        return dict(g0=self._vec(get("residual.0.gamma"), _cpad(cin)), c0=self._conv(get, "residual.2"),
                    g1=self._vec(get("residual.3.gamma"), _cpad(cout)), c1=self._conv(get, "residual.6"),
                    sc=self._conv(get, "shortcut") if cin != cout else None, cin=cin, cout=cout)

    def _attn(self, get, c):
        return dict(gamma=self._vec(get("norm.gamma"), _cpad(c)), qkv=self._conv(get, "to_qkv"), proj=self._conv(get, "proj"), c=c)

    def _conv2(self, scale):
        """conv2 with z / scale[1] + scale[0] folded in; cached per scale (the reference passes [mean, 1/std])."""
        # identity cache first: the reference hands the same two objects to every tile (wan_video_vae.py:643-692), so the host
        # sync of .tolist() is paid once per generation, not per tile
        ident = tuple((id(s), getattr(s, "_version", 0)) for s in scale)
        if self._conv2_ident is not None and self._conv2_ident[0] == ident:
            return self._conv2_ident[1]
        # per-channel tensors or plain scalars (VideoVAE_.decode accepts both, wan_video_vae.py:556-561): expand to z_dim
        mean, inv = (torch.as_tensor(s).float().reshape(-1).cpu().expand(self.z_dim).contiguous()
                     if torch.as_tensor(s).numel() == 1 else torch.as_tensor(s).float().reshape(-1).cpu() for s in scale)
        key = (tuple(mean.tolist()), tuple(inv.tolist()))
        if key not in self._conv2_cache:
            w = self._w2 / inv[None, :]
            b = self._w2 @ mean + self._b2
            self._conv2_cache = {key: self._pack(w, b, _cpad(self.z_dim), _cpad(self.z_dim))}
        self._conv2_ident = (ident, self._conv2_cache[key], tuple(scale))      # keeps the scale objects alive: ids stay unique
        return self._conv2_cache[key]

    # ------------------------------------------------------------------------------------------------ blocks
    def _res_apply(self, x, T, h, w, r):
        ops = self.ops
        sc = x if r["sc"] is None else ops.linear(x, r["sc"])
        y = self._conv_apply(ops.chan_rmsnorm_silu(x, r["g0"], r["cin"]), T, h, w, r["c0"], kt=3)
        return self._conv_apply(ops.chan_rmsnorm_silu(y, r["g1"], r["cout"]), T, h, w, r["c1"], kt=3, res=sc)

    def _attn_apply(self, x, T, h, w, a):
        """AttentionBlock (wan_video_vae.py:252-273): per frame, softmax(q k^T / sqrt(C)) v with a single head of width C."""
        ops = self.ops
        hw, C = h * w, a["c"]
        cp = _cpad(C)
        qkv = ops.linear(ops.chan_rmsnorm_silu(x, a["gamma"], C, silu=False), a["qkv"])          # [T*hw, 3C]
        hwp = _cpad(hw)
        o = ops.empty(T * hw, cp)
        for t in range(T):
            rows = qkv[t * hw:(t + 1) * hw]
            q, k, v = rows[:, :C], rows[:, C:2 * C], rows[:, 2 * C:3 * C]
            s = ops.linear(q.contiguous(), Linear(k.contiguous(), None), out_f32=True)             # [hw, hw]
            p = ops.softmax_rows(s, C ** -0.5, hwp)                                                  # [hw, hwp], zero padded
            vt = ops.empty(C, hwp)
            vt.zero_()
            vt[:, :hw] = v.t()
            ops.linear(p, Linear(vt, None), out=o[t * hw:(t + 1) * hw])
        return ops.linear(o, a["proj"], res=x)

    def _resample_apply(self, x, T, h, w, u):
        ops = self.ops
        hw, C = h * w, x.shape[1]
        if u["time"] is not None and T > 1:                       # upsample3d: frame 0 passes, frames 1.. are doubled in time
            y = self._conv_apply(x[hw:], T - 1, h, w, u["time"], kt=3, kh=1, kw=1)
            x = torch.cat([x[:hw], ops.unfold_time2(y, T - 1, hw, C)], dim=0)
            T = 2 * T - 1
        # nearest-exact x2 up-sampling folded into the gather of the 3x3 convolution
        return self._conv_apply(x, T, h, w, u["conv"], up=2), T, 2 * h, 2 * w

    # ------------------------------------------------------------------------------------------------ entry
    @torch.no_grad()
    def decode(self, z, scale=None):
        """z [1, z_dim, T, h, w] -> video [1, 3, 4T-3, 8h, 8w] in z.dtype (VideoVAE_.decode(z, scale), not clamped)."""
        ops = self.ops
        assert z.dim() == 5 and z.shape[0] == 1 and z.shape[1] == self.z_dim
        _, _, T, h, w = z.shape
        zin = ops.to_act(torch.zeros(T * h * w, _cpad(self.z_dim)))
        zin[:, :self.z_dim] = ops.to_act(z[0].permute(1, 2, 3, 0).reshape(T * h * w, self.z_dim))
        x = ops.linear(zin, self._conv2(self.default_scale if scale is None else scale))
        x = self._conv_apply(x, T, h, w, self.conv1, kt=3)
        x = self._res_apply(x, T, h, w, self.mid[0])
        x = self._attn_apply(x, T, h, w, self.mid[1])
        x = self._res_apply(x, T, h, w, self.mid[2])
        for blocks, up in self.stages:
            for r in blocks:
                x = self._res_apply(x, T, h, w, r)
            if up is not None:
                x, T, h, w = self._resample_apply(x, T, h, w, up)
        x = ops.chan_rmsnorm_silu(x, self.head_gamma, self.head_c)
        y = self._conv_apply(x, T, h, w, self.head, kt=3, out_f32=True)                 # [T*h*w, 3]
        return y.view(T, h, w, 3).permute(3, 0, 1, 2)[None].to(z.dtype)
