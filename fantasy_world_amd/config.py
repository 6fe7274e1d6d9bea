"""Static shape description of the FantasyWorld fusion model (Wan2.1 flavour).

The numbers mirror what the reference hard-codes or reads from the checkpoint config table:
DiT width 5120 / 40 heads / FFN 13824 (FantasyWorld/diffsynth_wan21/models/wan_video_dit.py:657-847 config table,
FantasyWorld/fusion/model_wan21.py:170), VGGT width 1024 / 16 heads / MLP 4096
(FantasyWorld/vggt/models/aggregator.py:52-70,164), bicross 1152 = 12 x 96 (model_wan21.py:63-64),
start_index 16, cross_attention_list range(24) (inference_wan21.py:204-212), camera adapter on DiT blocks <= 24
(wan_video_dit.py:515).
"""
from dataclasses import dataclass, field
from typing import List


@dataclass
class FWConfig:
    # WanDiT
    dim: int = 5120
    in_dim: int = 36
    ffn_dim: int = 13824
    out_dim: int = 16
    text_dim: int = 4096
    freq_dim: int = 256
    eps: float = 1e-6
    num_heads: int = 40
    num_layers: int = 40
    has_image_input: bool = True
    clip_dim: int = 1280
    clip_tokens: int = 257
    # fusion
    start_index: int = 16
    cross_attention_list: List[int] = field(default_factory=lambda: list(range(24)))
    bicross_dim: int = 1152
    bicross_heads: int = 12
    # VGGT aggregator
    vggt_dim: int = 1024
    vggt_heads: int = 16
    vggt_mlp: int = 4096
    vggt_eps: float = 1e-5
    n_special: int = 5           # 1 camera + 4 register tokens per frame
    vggt_rope_freq: float = 100.0
    # camera adapter (Wan2.1)
    camera_adapter: bool = True
    plucker_dim: int = 2048
    adapter_hidden: int = 1024   # min(hidden, context)//2
    adapter_reduced: int = 409   # context_dim // 5
    adapter_max_block: int = 24  # processors exist on DiT blocks 0..24
    # Wan2.2-Fun-A14B-Control-Camera: camera conditioning through a conv adapter added to the patch embedding
    # (diffsynth_wan22/models/wan_video_dit.py:842-858 config, :385-396 patchify), no CLIP context, no per-block adapter
    control_adapter: bool = False
    control_in_dim: int = 24

    @property
    def head_dim(self):
        return self.dim // self.num_heads

    @property
    def n_irg(self):
        return self.num_layers - self.start_index

    def dit_prefix(self, b: int) -> str:
        """Reference parameter-name prefix of DiT block b (moved into an IRGBlock when b-start_index is in the list)."""
        j = b - self.start_index
        if j >= 0 and j in self.cross_attention_list:
            return f"IRGBlock.{self.cross_attention_list.index(j)}.x_dit."
        return f"pipe.dit.blocks.{b}."

    def global_prefix(self, j: int) -> str:
        if j in self.cross_attention_list:
            return f"IRGBlock.{self.cross_attention_list.index(j)}.x_agg."
        return f"vggt.aggregator.global_blocks.{j}."

    def has_adapter(self, b: int) -> bool:
        return self.camera_adapter and b <= self.adapter_max_block


def wan21_14b() -> FWConfig:
    """BASELINE.json configs[1]/[2]: Wan2.1-I2V-14B-480P + IRG fusion + VGGT branch."""
    return FWConfig()


def wan22_a14b() -> FWConfig:
    """BASELINE.json configs[3]/[4]: one expert (high- or low-noise) of Wan2.2-Fun-A14B-Control-Camera + IRG fusion + VGGT."""
    return FWConfig(has_image_input=False, camera_adapter=False, control_adapter=True)


def plumbing22(num_layers: int = 2, start_index: int = 1, ffn_dim: int = 13824) -> FWConfig:
    """Reduced-depth Wan2.2 flavour (widths stay: they are hard-coded in the reference)."""
    n_irg = num_layers - start_index
    return FWConfig(num_layers=num_layers, start_index=start_index, ffn_dim=ffn_dim, has_image_input=False,
                    camera_adapter=False, control_adapter=True, cross_attention_list=list(range(n_irg)))


def plumbing(num_layers: int = 2, start_index: int = 1, ffn_dim: int = 13824) -> FWConfig:
    """BASELINE.json configs[0]: reduced-depth model (widths are hard-coded in the reference and stay)."""
    n_irg = num_layers - start_index
    return FWConfig(num_layers=num_layers, start_index=start_index, ffn_dim=ffn_dim,
                    cross_attention_list=list(range(n_irg)))


@dataclass
class HeadsConfig:
    """VGGT geometry heads (vggt/models/vggt.py:31-34 constructor arguments)."""
    dim_in: int = 2048                 # 2 * vggt_dim (frame | global intermediates concatenated)
    trunk_depth: int = 4               # camera_head.py:30
    cam_heads: int = 16
    cam_mlp_ratio: int = 4
    features: int = 256                # dpt_head.py:43
    out_channels: List[int] = field(default_factory=lambda: [256, 512, 1024, 1024])
    layer_idx: List[int] = field(default_factory=lambda: [23, 17, 11, 7])
    dpt_patch: int = 16                # vggt.py:26 DPT_patch_size
    depth_out: int = 2                 # depth + confidence (vggt.py:32)
    point_out: int = 4                 # xyz + confidence (vggt.py:33)

    @staticmethod
    def small():
        """Reduced widths for CPU tests (same structure; the reference modules take these as constructor arguments)."""
        return HeadsConfig(dim_in=128, trunk_depth=2, cam_heads=2, features=64, out_channels=[64, 64, 128, 128],
                           layer_idx=[3, 2, 1, 0], dpt_patch=4)

    @staticmethod
    def e2e_small():
        """Heads that hang on a reduced-depth fusion model (2 IRG layers, VGGT width 1024): true token width, narrow DPT."""
        return HeadsConfig(dim_in=2048, trunk_depth=1, cam_heads=16, features=64, out_channels=[64, 64, 128, 128],
                           layer_idx=[1, 0, 1, 0], dpt_patch=4)
