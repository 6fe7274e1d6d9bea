"""Host-side rotary tables (built once per latent grid, fp64 math, stored as fp32 (cos, sin) pairs).

The reference rebuilds three complex128 tables on the host and copies them to the device on EVERY forward
(FantasyWorld/fusion/model_wan21.py:132-147); here they are step-invariant device constants consumed by
fw_qk_prep.  Layout of every table: [tokens][head_dim/2][2] = (cos, sin) of the angle of rotary pair i.
"""
import torch


def _freqs_1d(dim, end, theta=10000.0):
    # FantasyWorld/diffsynth_wan21/models/wan_video_dit.py:88-94 (precompute_freqs_cis): angle[pos, i]
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].double() / dim))
    return torch.outer(torch.arange(end).double(), freqs)


def rope3d_angles(head_dim, f, h, w):
    """[f*h*w, head_dim/2] fp64 angles; pair order = [frame pairs | row pairs | col pairs]
    (wan_video_dit.py:80-86 split dim-2*(dim//3), dim//3, dim//3; model_wan21.py:132-136 expansion)."""
    df = head_dim - 2 * (head_dim // 3)
    dh = head_dim // 3
    af = _freqs_1d(df, f)
    ah = _freqs_1d(dh, h)
    aw = _freqs_1d(dh, w)
    ang = torch.cat([
        af.view(f, 1, 1, -1).expand(f, h, w, -1),
        ah.view(1, h, 1, -1).expand(f, h, w, -1),
        aw.view(1, 1, w, -1).expand(f, h, w, -1),
    ], dim=-1)
    return ang.reshape(f * h * w, head_dim // 2)


def _to_table(ang):
    return torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).to(torch.float32).contiguous()


def rope3d_table(head_dim, f, h, w):
    return _to_table(rope3d_angles(head_dim, f, h, w))


def rope3d_table_with_extra(head_dim, f, h, w, n_extra):
    """Per frame: n_extra identity rows (special tokens are not rotated) then the h*w patch rows
    (wan_video_dit.py:105-132 build_freqs_3d_with_extra_cis)."""
    ang = rope3d_angles(head_dim, f, h, w).view(f, h * w, head_dim // 2)
    extra = torch.zeros(f, n_extra, head_dim // 2, dtype=ang.dtype)
    return _to_table(torch.cat([extra, ang], dim=1).reshape(f * (n_extra + h * w), head_dim // 2))


def rope2d_table(head_dim, h, w, n_special, base=100.0):
    """VGGT 2-D rotary table for ONE frame: [n_special + h*w, head_dim/2, 2].

    Positions: special tokens (0,0), patch (y,x) -> (y+1, x+1) (vggt/models/aggregator.py:276-280).  Pair index
    i < head_dim/4 rotates (i, i+head_dim/4) of the y half by pos_y * inv_freq[i]; the next head_dim/4 pairs do the
    same on the x half (vggt/layers/rope.py:82-131,154-188).  Angles are formed in fp32 exactly as the reference's
    fp32 path does (positions.float() * inv_freq.float()).
    """
    half = head_dim // 2
    exponents = torch.arange(0, half, 2).float() / half
    inv_freq = 1.0 / (base ** exponents)                           # [head_dim/4] fp32
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    pos = torch.stack([ys.reshape(-1), xs.reshape(-1)], dim=-1) + 1  # [hw, 2]
    pos = torch.cat([torch.zeros(n_special, 2, dtype=pos.dtype), pos], dim=0).float()
    ang_y = pos[:, 0:1] * inv_freq[None, :]
    ang_x = pos[:, 1:2] * inv_freq[None, :]
    ang = torch.cat([ang_y, ang_x], dim=-1)                        # [P, head_dim/2] fp32
    return torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).to(torch.float32).contiguous()
