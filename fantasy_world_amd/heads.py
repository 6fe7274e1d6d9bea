"""VGGT geometry heads on the engine's op set: the once-per-generation tail of joint_forward.

Mirrors `VGGT._head_predction` (FantasyWorld/vggt/models/vggt.py:134-154): CameraHead (vggt/heads/camera_head.py:76-145)
-> pose_enc, DPTHead_3D_Causal (vggt/heads/dpt_head.py:133-320) x2 -> depth / world points + confidences, with the
temporal up-sampler WanVAE_(location="DPT").decode (wan/modules/vae_modified.py:443-476) in between.

Layout: every feature map is a channels-last matrix [frames*H*W, C] (bf16, C padded to a multiple of 64 with zero
channels), so a 1x1 convolution is a plain GEMM, a k x k (x k) convolution is a gather (`im2col`, tap-major columns) followed
by the same GEMM, and a ConvTranspose2d with kernel = stride is a GEMM followed by a depth-to-space shuffle.  All matrix
work goes through fw_gemm_bf16 with the bias / ReLU / residual fused in its epilogue.

The reference decodes time frame by frame through a convolution cache; every CausalConv3d sees the two previous frames of
its own input sequence (zeros before the start), and the up-sampler passes frame 0 through and convolves frames 1.. with zero
history -- so whole-sequence causal convolutions give the same numbers (oracle/fw_heads_oracle.py pins that against the
chunked reference).  Frames are chunked here only to bound the size of the gathered matrices.
"""
import torch

from .config import HeadsConfig
from .convnet import ConvNetBase, _cpad


def uv_pos_embed(C, ph, pw, aspect, ratio=0.1, omega_0=100.0):
    """fp32 [ph*pw, C]: dpt_head.py:262-283 with heads/utils.py:11-109 (create_uv_grid -> sin/cos of u and v at
    omega_0^(-i/(C/4)), scaled by 0.1).  Input independent; built once per shape on the host."""
    diag = (aspect ** 2 + 1.0) ** 0.5
    sx, sy = aspect / diag, 1.0 / diag
    xs = torch.linspace(-sx * (pw - 1) / pw, sx * (pw - 1) / pw, steps=pw, dtype=torch.float32)
    ys = torch.linspace(-sy * (ph - 1) / ph, sy * (ph - 1) / ph, steps=ph, dtype=torch.float32)
    uu, vv = torch.meshgrid(xs, ys, indexing="xy")

    def sincos(dim, pos):
        omega = 1.0 / omega_0 ** (torch.arange(dim // 2, dtype=torch.double) / (dim / 2.0))
        out = pos.reshape(-1, 1).double() * omega[None]
        return torch.cat([torch.sin(out), torch.cos(out)], dim=1).float()

    return torch.cat([sincos(C // 2, uu), sincos(C // 2, vv)], dim=-1) * ratio


class _DPT:
    pass


class GeometryHeads(ConvNetBase):
    """hc: HeadsConfig; get(name) -> tensor with the reference's parameter names ("vggt.camera_head..." etc.); ops: HipOps."""

    def __init__(self, hc: HeadsConfig, get, ops, max_col_bytes=4 << 30, frames_chunk=16):
        self.hc, self.ops = hc, ops
        self.max_col_bytes = max_col_bytes
        self.frames_chunk = frames_chunk          # dpt_head.py:138 frames_chunk_size_second
        self._pos_cache = {}
        self._pack_camera(get)
        self.depth = self._pack_dpt(get, "vggt.depth_head.", hc.depth_out, "exp")
        self.point = self._pack_dpt(get, "vggt.point_head.", hc.point_out, "inv_log")

    # ------------------------------------------------------------------------------------------------ weight packing
    def _pack_camera(self, get):
        hc, p, ops = self.hc, "vggt.camera_head.", self.ops
        C = hc.dim_in
        self.cam_blocks = []
        for b in range(hc.trunk_depth):
            q = f"{p}trunk.{b}."
            self.cam_blocks.append(dict(
                n1=(self._vec(get(q + "norm1.weight")), self._vec(get(q + "norm1.bias"))),
                qkv=self._lin(get, q + "attn.qkv"), proj=self._lin(get, q + "attn.proj"), ls1=self._vec(get(q + "ls1.gamma")),
                n2=(self._vec(get(q + "norm2.weight")), self._vec(get(q + "norm2.bias"))),
                fc1=self._lin(get, q + "mlp.fc1"), fc2=self._lin(get, q + "mlp.fc2"), ls2=self._vec(get(q + "ls2.gamma"))))
        self.cam_token_norm = (self._vec(get(p + "token_norm.weight")), self._vec(get(p + "token_norm.bias")))
        self.cam_trunk_norm = (self._vec(get(p + "trunk_norm.weight")), self._vec(get(p + "trunk_norm.bias")))
        self.cam_empty = get(p + "empty_pose_tokens").float().reshape(1, 9)
        self.cam_embed = self._lin(get, p + "embed_pose")                       # K = 9 -> 64
        self.cam_mod = self._lin(get, p + "poseLN_modulation.1")
        self.cam_up = self._lin(get, p + "camera_time_upsample.expand_channels")
        self.cam_fc1 = self._lin(get, p + "pose_branch.fc1")
        self.cam_fc2 = self._lin(get, p + "pose_branch.fc2")
        assert C % hc.cam_heads == 0 and C // hc.cam_heads in (64, 96, 128), "camera trunk head_dim must be 64/96/128"

    def _pack_dpt(self, get, pre, odim, activation):
        hc = self.hc
        d = _DPT()
        d.activation, d.odim = activation, odim
        d.norm = (self._vec(get(pre + "norm.weight")), self._vec(get(pre + "norm.bias")))
        oc = hc.out_channels
        assert all(c % 64 == 0 for c in oc) and hc.features % 64 == 0, "DPT widths must be multiples of 64"
        d.proj = [self._conv(get, pre + f"projects.{i}") for i in range(4)]
        d.resize = [self._convT(get, pre + "resize_layers.0", 4), self._convT(get, pre + "resize_layers.1", 2), None,
                    self._conv(get, pre + "resize_layers.3")]
        d.temporal = []
        for i, c in enumerate(oc):
            t = pre + f"temporal_upsamplers.{i}."
            d.temporal.append(dict(
                conv2=self._conv(get, t + "conv2"),
                up=[self._conv(get, t + f"decoder.upsamples.{u}.time_conv") for u in (0, 2)],
                gamma=[self._vec(get(t + f"decoder.upsamples.{u}.residual.0.gamma"), _cpad(c)) for u in (1, 3)],
                rb=[self._conv(get, t + f"decoder.upsamples.{u}.residual.2") for u in (1, 3)]))
        sc = pre + "scratch."
        d.rn = [self._conv(get, sc + f"layer{i + 1}_rn", bias=False) for i in range(4)]
        d.fusion = {}
        for r in (1, 2, 3, 4):
            q = sc + f"refinenet{r}."
            f = dict(out=self._conv(get, q + "out_conv"))
            for u in ((1, 2) if r != 4 else (2,)):
                f[f"rcu{u}"] = (self._conv(get, q + f"resConfUnit{u}.conv1"), self._conv(get, q + f"resConfUnit{u}.conv2"))
            d.fusion[r] = f
        d.oc1 = self._conv(get, sc + "output_conv1")
        d.oc2a = self._conv(get, sc + "output_conv2.0")
        d.oc2b = self._conv(get, sc + "output_conv2.2", pad_out=False)
        return d

    # ------------------------------------------------------------------------------------------------ building blocks
    def _pos(self, C_true, ph, pw, aspect, C_pad):
        key = (C_true, ph, pw, round(aspect, 9), C_pad)
        if key not in self._pos_cache:
            t = torch.zeros(ph * pw, C_pad, dtype=torch.float32)
            t[:, :C_true] = uv_pos_embed(C_true, ph, pw, aspect)
            self._pos_cache[key] = self.ops.to_f32(t)
        return self._pos_cache[key]

    def _upsample3d(self, x, T, h, w, lin):
        """Resample 'upsample3d' (vae_modified.py:87-130): frame 0 unchanged; frames 1.. through CausalConv3d(C, 2C, (3,1,1))
        with zero history, every output split into two frames."""
        if T == 1:
            return x, 1
        hw, C = h * w, x.shape[1]
        y = self._conv_apply(x[hw:], T - 1, h, w, lin, kt=3, kh=1, kw=1)
        up = self.ops.unfold_time2(y, T - 1, hw, C)
        return torch.cat([x[:hw], up], dim=0), 2 * T - 1

    def _temporal_decode(self, x, S, h, w, tw, c_true):
        x = self.ops.linear(x, tw["conv2"])
        T = S
        for u in range(2):
            x, T = self._upsample3d(x, T, h, w, tw["up"][u])
            hn = self.ops.chan_rmsnorm_silu(x, tw["gamma"][u], c_true)          # ResidualBlock_Half, vae_modified.py:193-226
            x = self._conv_apply(hn, T, h, w, tw["rb"][u], kt=3, res=x)
        return x, T

    def _rcu(self, x, T, h, w, convs):
        """ResidualConvUnit (dpt_head.py:400-457) on an input that already went through the in-place ReLU."""
        h1 = self._conv_apply(x, T, h, w, convs[0], act="relu")
        return self._conv_apply(h1, T, h, w, convs[1], res=x)

    def _fusion(self, f, x, skip, T, h, w, size):
        """FeatureFusionBlock.forward (dpt_head.py:508-536).  `skip` and (for the residual-free block) `x` arrive ReLU'd."""
        ops = self.ops
        if skip is not None:
            x = ops.add_act(x, self._rcu(skip, T, h, w, f["rcu1"]), relu=True)
        x = self._rcu(x, T, h, w, f["rcu2"])
        x = ops.resize_bilinear(x, T, h, w, size[0], size[1])
        return ops.linear(x, f["out"])

    # ------------------------------------------------------------------------------------------------ DPT head
    def _dpt(self, d, output_list, S, ph, pw, psi):
        hc, ops = self.hc, self.ops
        H, Wd = ph * hc.dpt_patch, pw * hc.dpt_patch
        aspect = Wd / H
        feats, sizes = [], []
        T = (S - 1) * 4 + 1
        for i, layer in enumerate(hc.layer_idx):
            tok = output_list[layer]
            tok = tok.reshape(S, -1, tok.shape[-1])[:, psi:].reshape(S * ph * pw, -1).contiguous()
            x = ops.layernorm(ops.to_f32(tok), w=d.norm[0], b=d.norm[1], eps=1e-5)
            x = ops.linear(x, d.proj[i])
            ops.add_table(x, self._pos(hc.out_channels[i], ph, pw, aspect, x.shape[1]))
            h, w = ph, pw
            if i == 0 or i == 1:
                k = 4 if i == 0 else 2
                x = ops.depth_to_space(ops.linear(x, d.resize[i]), S, ph, pw, k, x.shape[1])
                h, w = ph * k, pw * k
            elif i == 3:
                x = self._conv_apply(x, S, ph, pw, d.resize[3], sh=2, sw=2)
                h, w = (ph + 2 - 3) // 2 + 1, (pw + 2 - 3) // 2 + 1
            x, Tn = self._temporal_decode(x, S, h, w, d.temporal[i], hc.out_channels[i])
            assert Tn == T
            feats.append(x)
            sizes.append((h, w))
        pts_out = ops.empty(T * H * Wd, d.odim - 1, dtype=torch.float32)
        conf_out = ops.empty(T * H * Wd, dtype=torch.float32)
        for t0 in range(0, T, self.frames_chunk):
            n = min(self.frames_chunk, T - t0)
            sub = [f[t0 * hh * ww:(t0 + n) * hh * ww] for f, (hh, ww) in zip(feats, sizes)]
            # layerN_rn 3x3 convolutions; their outputs only ever enter ResidualConvUnits, whose in-place ReLU rewrites them
            l = [self._conv_apply(s, n, hh, ww, d.rn[i], act="relu") for i, (s, (hh, ww)) in enumerate(zip(sub, sizes))]
            out = self._fusion(d.fusion[4], l[3], None, n, *sizes[3], sizes[2])
            out = self._fusion(d.fusion[3], out, l[2], n, *sizes[2], sizes[1])
            out = self._fusion(d.fusion[2], out, l[1], n, *sizes[1], sizes[0])
            h1, w1 = sizes[0]
            out = self._fusion(d.fusion[1], out, l[0], n, h1, w1, (2 * h1, 2 * w1))
            out = self._conv_apply(out, n, 2 * h1, 2 * w1, d.oc1)
            out = ops.resize_bilinear(out, n, 2 * h1, 2 * w1, H, Wd)
            ops.add_table(out, self._pos(hc.features // 2, H, Wd, aspect, out.shape[1]))
            out = self._conv_apply(out, n, H, Wd, d.oc2a, act="relu")
            y = ops.linear(out, d.oc2b, out_f32=True)
            pts, conf = ops.head_activation(y, d.activation)
            sl = slice(t0 * H * Wd, (t0 + n) * H * Wd)
            pts_out[sl].copy_(pts)
            conf_out[sl].copy_(conf)
        return pts_out.view(1, T, H, Wd, d.odim - 1), conf_out.view(1, T, H, Wd)

    # ------------------------------------------------------------------------------------------------ camera head
    def _camera(self, tokens_last, num_iterations=4):
        hc, ops = self.hc, self.ops
        C, heads = hc.dim_in, hc.cam_heads
        hd = C // heads
        tok = tokens_last.reshape(-1, tokens_last.shape[-2], C)
        pose = ops.to_f32(tok[:, 0])                                                    # [S, C] camera token of every latent frame
        S = pose.shape[0]
        first = ops.layernorm(pose[:1].contiguous(), w=self.cam_token_norm[0], b=self.cam_token_norm[1], eps=1e-5)
        if S > 1:
            y = ops.linear(ops.cast_act(pose[1:].contiguous()), self.cam_up)             # Conv1d(C, 4C, 1)
            # vae_modified.py:565-572: [4C, N] reinterpreted as [C, 4N] -- token j*N + n takes channels 4c + j of token n
            up = y.view(S - 1, C, 4).permute(2, 0, 1).reshape(4 * (S - 1), C)
            x0 = torch.cat([first.float(), up.float()], dim=0).contiguous()             # the up-sampled tokens are not normalised
        else:
            x0 = first.float()
        T = x0.shape[0]
        pred = None
        for _ in range(num_iterations):
            inp = ops.to_f32(torch.zeros(T, self.cam_embed.K))                           # embed_pose input, K padded 9 -> 64
            inp[:, :9] = ops.to_f32(self.cam_empty).expand(T, 9) if pred is None else pred
            mod = ops.linear(ops.linear(ops.cast_act(inp), self.cam_embed, act="silu"), self.cam_mod, out_f32=True)
            x = ops.adaln_rows(x0, mod)                                                  # fp32 residual stream
            for blk in self.cam_blocks:
                h = ops.layernorm(x, w=blk["n1"][0], b=blk["n1"][1], eps=1e-5)
                qkv = ops.linear(h, blk["qkv"])
                ops.qk_prep(qkv[:, :C], heads, hd, out_scale=ops.q_scale(hd))
                a = ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], heads, hd, q_prescaled=True)
                x = ops.linear(a, blk["proj"], g1=blk["ls1"], res=x, out_f32=True)
                h = ops.layernorm(x, w=blk["n2"][0], b=blk["n2"][1], eps=1e-5)
                x = ops.linear(ops.linear(h, blk["fc1"], act="gelu_erf"), blk["fc2"], g1=blk["ls2"], res=x, out_f32=True)
            h = ops.layernorm(x, w=self.cam_trunk_norm[0], b=self.cam_trunk_norm[1], eps=1e-5)
            pred = ops.linear(ops.linear(h, self.cam_fc1, act="gelu_erf"), self.cam_fc2, res=pred, out_f32=True)
        return ops.head_activation(pred, "pose").view(1, T, 9)

    # ------------------------------------------------------------------------------------------------ entry
    @torch.no_grad()
    def predict(self, output_list, S, ph, pw, patch_start_idx=5):
        """output_list: layer -> fp32 [1, S, P, dim_in] (the layers HeadsConfig.layer_idx names and the last one).
        Returns the reference's prediction dict (vggt.py:134-154), fp32."""
        last = output_list[max(output_list.keys())]
        pred = {"pose_enc": self._camera(last)}
        pred["depth"], pred["depth_conf"] = self._dpt(self.depth, output_list, S, ph, pw, patch_start_idx)
        pred["world_points"], pred["world_points_conf"] = self._dpt(self.point, output_list, S, ph, pw, patch_start_idx)
        return pred
