"""fantasy_world_amd: MI355X-native (gfx950) execution of FantasyWorld's per-step denoising forward
(Fantasy-AMAP/fantasy-world, FantasyWorldFusionModel.joint_forward) behind the reference's own call surface."""
from .config import FWConfig, HeadsConfig, wan21_14b, wan22_a14b, plumbing, plumbing22  # noqa: F401
from .install import install, uninstall, install_flash_attention, install_vae  # noqa: F401
from .hooks import install_bicross_attention, install_layernorm_kernel, HipLayerNorm, HipLinear  # noqa: F401
from .blocks import install_blocks, install_dit_block, install_vggt_block, install_irg_block  # noqa: F401

__all__ = ["FWConfig", "HeadsConfig", "wan21_14b", "wan22_a14b", "plumbing", "plumbing22", "install", "uninstall",
           "install_flash_attention", "install_vae", "install_bicross_attention", "install_layernorm_kernel", "HipLayerNorm",
           "HipLinear", "install_blocks", "install_dit_block", "install_vggt_block", "install_irg_block"]
