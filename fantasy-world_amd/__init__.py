"""fantasy-world_amd: MI355X-native (gfx950) execution of FantasyWorld's per-step denoising forward.

The directory name carries a hyphen (it mirrors the reference repo's name); import it as `fantasy_world_amd`
(the tiny shim package of that name at the repo root redirects here).
"""
from .config import FWConfig, HeadsConfig, wan21_14b, wan22_a14b, plumbing, plumbing22  # noqa: F401
from .install import install, uninstall, install_flash_attention, install_vae  # noqa: F401

__all__ = ["FWConfig", "HeadsConfig", "wan21_14b", "wan22_a14b", "plumbing", "plumbing22", "install", "uninstall",
           "install_flash_attention", "install_vae"]
