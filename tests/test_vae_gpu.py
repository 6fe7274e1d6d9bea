"""-m gpu: the Wan video VAE decoder through libfw_mi355x.so.

  * the kernels added for it against the torch statement of the same op: the gather that folds the x2 nearest-neighbour
    up-sampling into a 3x3 convolution (bit-exact), the bare channel RMS norm and the row softmax (one bf16 rounding, 4e-3);
  * the decoded frames of fantasy_world_amd.vae_decoder on HipOps against the golden output of the REAL reference
    (VideoVAE_.decode, fp32 CPU) at its real widths.  Tolerance 2e-2 rel-L2: bf16 is stored between 34 chained 3-D convolutions
    (the reference decodes in bf16 too); the same host code on the torch ops with that rounding emulated measures 1.05e-2;
  * one full reference tile (21 x 34 x 34 latents -> 81 x 272 x 272 frames): shape, finiteness, frame chunking invisible.
"""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu

VAE_TOL = 2e-2


@pytest.fixture(scope="module")
def ops():
    from fantasy_world_amd.hip_ops import HipOps
    return HipOps("cuda:0")


@pytest.fixture(scope="module")
def ref():
    from oracle.ref_ops import TorchRefOps
    return TorchRefOps()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).float()


@pytest.mark.parametrize("T,H,W,C,kt", [(2, 3, 5, 64, 1), (3, 4, 4, 72, 1), (4, 2, 3, 64, 3)])
def test_upsampled_gather_is_exact(ops, ref, T, H, W, C, kt):
    x = rnd(T * H * W, C, seed=1)
    want = ref.im2col(x, T, H, W, kt, 3, 3, up=2)
    got = ops.im2col(x.to(torch.bfloat16).cuda(), T, H, W, kt, 3, 3, up=2)
    assert got.shape == (T * 4 * H * W, kt * 9 * C) and torch.equal(got.float().cpu(), want)


def test_upsampled_gather_then_gemm_is_upsample_plus_conv(ops, parity, request):
    """= nn.Upsample(scale_factor=2, mode='nearest-exact') followed by Conv2d 3x3 (Resample, wan_video_vae.py:92-99)."""
    import torch.nn.functional as F
    T, H, W, C, N = 2, 5, 7, 64, 128
    x, w, b = rnd(T * H * W, C, seed=2), rnd(N, C, 3, 3, seed=3, scale=(9 * C) ** -0.5), rnd(N, seed=4, scale=0.1)
    img = F.interpolate(x.view(T, H, W, C).permute(0, 3, 1, 2), scale_factor=(2.0, 2.0), mode="nearest-exact")
    want = F.conv2d(img, w, b, padding=1).permute(0, 2, 3, 1).reshape(T * 4 * H * W, N)
    lin = ops.pack_linear(w.permute(0, 2, 3, 1).reshape(N, 9 * C), b)
    got = ops.linear(ops.im2col(x.to(torch.bfloat16).cuda(), T, H, W, 1, 3, 3, up=2), lin, out_f32=True)
    parity.check(f"op/{request.node.name}/0", rel_l2(got, want), 1e-3)


def test_bare_channel_rms_norm_and_row_softmax(ops, ref):
    x = rnd(100, 384, seed=5, scale=2.0)
    g = 1 + 0.1 * rnd(384, seed=6)
    assert rel_l2(ops.chan_rmsnorm_silu(x.to(torch.bfloat16).cuda(), g.cuda(), 384, silu=False).float(),
                  ref.chan_rmsnorm_silu(x, g, 384, silu=False)) < 4e-3
    for rows, cols in [(37, 24), (100, 1156), (5, 6240)]:
        s = rnd(rows, cols, seed=7, scale=30.0)
        pad = (cols + 63) // 64 * 64
        want = ref.softmax_rows(s, 384 ** -0.5, pad)
        got = ops.softmax_rows(s.cuda(), 384 ** -0.5, pad)
        assert got.shape == (rows, pad) and rel_l2(got.float(), want) < 4e-3
        assert (got[:, cols:] == 0).all() and (got.float().sum(dim=1) - 1).abs().max() < 2e-2


def test_vae_decoder_matches_reference_golden(vae_case, ops):
    from fantasy_world_amd.vae_decoder import VaeDecoder
    c = vae_case
    dec = VaeDecoder(c.weights.__getitem__, ops)
    got = dec.decode(c.latents.cuda())
    torch.cuda.synchronize()
    want = c.golden["video"]
    err = rel_l2(got, want)
    print(c.name, f"{err:.2e}")
    assert got.shape == want.shape and got.dtype == torch.float32 and torch.isfinite(got).all()
    assert err < VAE_TOL
    scale = [dec.default_scale[0].clone(), dec.default_scale[1].clone()]                          # the call the reference makes
    assert torch.equal(dec.decode(c.latents.cuda(), scale), got)


def test_vae_decoder_reference_tile(ops):
    """One tile of WanVideoVAE.tiled_decode (tile_size 34 x 34, wan_video_vae.py:776): 21 latent frames -> 81 x 272 x 272."""
    from fantasy_world_amd import synth
    from fantasy_world_amd.vae_decoder import VaeDecoder
    W = synth.make_vae_decoder_weights(device="cuda")
    z = synth.make_latents(21, 34, 34, device="cuda").bfloat16()
    a = VaeDecoder(W.__getitem__, ops).decode(z)
    b = VaeDecoder(W.__getitem__, ops, max_col_bytes=256 << 20).decode(z)
    torch.cuda.synchronize()
    assert a.shape == (1, 3, 81, 272, 272) and a.dtype == torch.bfloat16 and torch.isfinite(a.float()).all()
    assert torch.equal(a, b)
