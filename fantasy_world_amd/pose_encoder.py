"""CameraPoseEncoder on the engine's op set: the once-per-generation producer of `plucker_fea`.

Mirrors FantasyWorld/diffsynth_wan21/models/pose_adaptor_ac3d.py:83-118 ('adaln' injection), which
`CameraConditionModel.get_pose_fea` (camera_control.py:233-234) calls before the sampling loop (fusion/model_wan21.py:271):
Pluecker embedding [1, 81, H, W, 6] -> PixelUnshuffle(8) -> (1x1 conv, GroupNorm(2)) x2 + ReLU -> temporal pooling 81 -> 41 ->
1x1 conv, GroupNorm(2), ReLU -> 41 -> 21 -> Conv3d(768 -> 5120, kernel = stride = (1,2,2)) -> Linear, LayerNorm, GELU, Linear,
LayerNorm -> [1, 21*(H/16)*(W/16), 2048].  Channels-last rows throughout: the 1x1 convolutions and the patch embedding are
fw_gemm_bf16 calls, the rest are streaming kernels.
"""
import torch


def _cpad(c):
    return (c + 63) // 64 * 64


class PoseEncoder:
    def __init__(self, get, ops, pre="camera_condition.pose_encoder."):
        self.ops = ops
        g = lambda n: get(pre + n).detach().float()
        vec = lambda n: ops.to_f32(g(n).reshape(-1))

        def conv1x1(name):
            w = g(name + ".weight")
            return ops.pack_linear(w.reshape(w.shape[0], -1), g(name + ".bias"))

        self.c0, self.c2 = conv1x1("controlnet_encode_first.0"), conv1x1("controlnet_encode_first.2")
        self.gn1 = (vec("controlnet_encode_first.1.weight"), vec("controlnet_encode_first.1.bias"))
        self.gn3 = (vec("controlnet_encode_first.3.weight"), vec("controlnet_encode_first.3.bias"))
        self.s0 = conv1x1("controlnet_encode_second.0")
        self.gns = (vec("controlnet_encode_second.1.weight"), vec("controlnet_encode_second.1.bias"))
        w = g("patch_embedding.weight")                                             # [dim, C, 1, 2, 2]
        self.patch = ops.pack_linear(w.permute(0, 2, 3, 4, 1).reshape(w.shape[0], -1), g("patch_embedding.bias"))   # tap-major
        self.fc0 = ops.pack_linear(g("fc.0.weight"), g("fc.0.bias"))
        self.fc3 = ops.pack_linear(g("fc.3.weight"), g("fc.3.bias"))
        self.ln1 = (vec("fc.1.weight"), vec("fc.1.bias"))
        self.ln4 = (vec("fc.4.weight"), vec("fc.4.bias"))
        assert self.c0.K % 64 == 0 and self.s0.K % 64 == 0, "pose encoder widths must be multiples of 64"

    @torch.no_grad()
    def encode(self, plucker):
        """plucker [1, F, H, W, C] (any float dtype) -> plucker_fea [1, L, context_dim] in plucker.dtype."""
        ops = self.ops
        assert plucker.dim() == 5 and plucker.shape[0] == 1, "the reference samples with batch 1"
        _, Fr, H, W, _ = plucker.shape
        assert H % 16 == 0 and W % 16 == 0, "PixelUnshuffle(8) followed by the (1,2,2) patch embedding"
        h, w = H // 8, W // 8
        x = ops.pixel_unshuffle_rows(plucker[0], 8)                                 # [F*h*w, 64*C]
        x = ops.group_norm_rows(ops.linear(x, self.c0), Fr, 2, *self.gn1)
        x = ops.group_norm_rows(ops.linear(x, self.c2), Fr, 2, *self.gn3, relu=True)
        x, f1 = ops.time_avg_pool(x, Fr, h * w)
        x = ops.group_norm_rows(ops.linear(x, self.s0), f1, 2, *self.gns, relu=True)
        x, f2 = ops.time_avg_pool(x, f1, h * w)
        x = ops.linear(ops.im2col(x, f2, h, w, 1, 2, 2, sh=2, sw=2, ph=0, pw=0), self.patch)   # Conv3d k = s = (1,2,2)
        x = ops.layernorm(ops.linear(x, self.fc0), w=self.ln1[0], b=self.ln1[1], eps=1e-5)
        x = ops.linear(ops.activation(x, "gelu_erf"), self.fc3)
        x = ops.layernorm(x, w=self.ln4[0], b=self.ln4[1], eps=1e-5)
        return x.view(1, x.shape[0], x.shape[1]).to(plucker.dtype)
