"""TEST INFRASTRUCTURE ONLY -- CPU restatement (PyTorch fp32) of FantasyWorld's CameraPoseEncoder.

`CameraConditionModel.get_pose_fea` (FantasyWorld/diffsynth_wan21/models/camera_control.py:233-234) runs it once per
generation, before the sampling loop (fusion/model_wan21.py:271): Pluecker embedding [1, 81, H, W, 6] -> plucker_fea
[1, 21*(H/16)*(W/16), 2048], the tensor every joint_forward call receives.  Functional restatement of
diffsynth_wan21/models/pose_adaptor_ac3d.py:8-118; pinned against the real module (tests/test_oracle_pin.py) and its
committed output (tests/golden/pose_*.pt).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import torch
import torch.nn.functional as F


def compress_time(x):
    # pose_adaptor_ac3d.py:61-76: x [F, C, h, w]; odd F keeps frame 0 and averages the remaining frames in pairs,
    # even F averages all pairs (avg_pool1d(kernel 2, stride 2): a trailing odd frame is dropped)
    n = x.shape[0]
    if n % 2 == 1:
        rest = x[1:]
        rest = (rest[0::2] + rest[1::2]) * 0.5 if rest.shape[0] > 0 else rest
        return torch.cat([x[:1], rest], dim=0)
    return (x[0:n - 1:2] + x[1:n:2]) * 0.5


def camera_pose_encoder(W, plucker, pre="camera_condition.pose_encoder."):
    """plucker [1, F, H, W, C] -> plucker_fea [1, F'' * (H/16) * (W/16), context_dim]  (pose_adaptor_ac3d.py:83-118, 'adaln')."""
    assert plucker.shape[0] == 1
    x = plucker[0].permute(0, 3, 1, 2)                                              # (b f) c h w
    x = F.pixel_unshuffle(x, 8)
    p = pre + "controlnet_encode_first."
    x = F.conv2d(x, W[p + "0.weight"], W[p + "0.bias"])
    x = F.group_norm(x, 2, W[p + "1.weight"], W[p + "1.bias"], 1e-5)
    x = F.conv2d(x, W[p + "2.weight"], W[p + "2.bias"])
    x = F.relu(F.group_norm(x, 2, W[p + "3.weight"], W[p + "3.bias"], 1e-5))
    x = compress_time(x)
    p = pre + "controlnet_encode_second."
    x = F.conv2d(x, W[p + "0.weight"], W[p + "0.bias"])
    x = F.relu(F.group_norm(x, 2, W[p + "1.weight"], W[p + "1.bias"], 1e-5))
    x = compress_time(x)
    x = x.permute(1, 0, 2, 3)[None]                                                 # b c f h w
    x = F.conv3d(x, W[pre + "patch_embedding.weight"], W[pre + "patch_embedding.bias"], stride=(1, 2, 2))
    x = x.flatten(2).transpose(1, 2)                                                # b (f h w) c
    p = pre + "fc."
    x = F.linear(x, W[p + "0.weight"], W[p + "0.bias"])
    x = F.gelu(F.layer_norm(x, (x.shape[-1],), W[p + "1.weight"], W[p + "1.bias"], 1e-5))
    x = F.linear(x, W[p + "3.weight"], W[p + "3.bias"])
    return F.layer_norm(x, (x.shape[-1],), W[p + "4.weight"], W[p + "4.bias"], 1e-5)
