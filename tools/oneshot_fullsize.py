"""The once-per-generation stages at BASELINE config-2 size on one MI355X: CameraPoseEncoder (81 x 480 x 832 Pluecker map ->
plucker_fea [1, 32760, 2048]) and the Wan VAE decoder (one 34 x 34 reference tile, and the untiled 21 x 60 x 104 latents ->
81 x 480 x 832 frames).  Wall time and peak memory; run under rocprofv3 (tools/gpu_oneshot_prof.sh) for the kernel breakdown."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fantasy_world_amd import synth                                 # noqa: E402
from fantasy_world_amd.hip_ops import HipOps                        # noqa: E402
from fantasy_world_amd.pose_encoder import PoseEncoder              # noqa: E402
from fantasy_world_amd.vae_decoder import VaeDecoder                # noqa: E402


def timed(tag, fn, reps=2):
    for rep in range(reps):
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.time()
        out = fn()
        torch.cuda.synchronize()
        print(f"{tag}: rep {rep} {1e3 * (time.time() - t0):.0f} ms, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, "
              f"out {tuple(out.shape)}", flush=True)
    assert torch.isfinite(out.float()).all()


def main():
    ops = HipOps("cuda:0")
    enc = PoseEncoder(synth.make_pose_encoder_weights(device="cuda").__getitem__, ops)
    pl = synth.make_plucker(81, 480, 832, device="cuda").bfloat16()
    timed("pose encoder 81x480x832", lambda: enc.encode(pl))
    dec = VaeDecoder(synth.make_vae_decoder_weights(device="cuda").__getitem__, ops)
    dec.implicit_conv = "--gather" not in sys.argv      # --gather: the older fw_im2col + fw_gemm_bf16 pair (A/B)
    print("convolutions:", "implicit GEMM" if dec.implicit_conv else "gather + GEMM", flush=True)
    tile = synth.make_latents(21, 34, 34, device="cuda").bfloat16()
    timed("vae decoder tile 21x34x34", lambda: dec.decode(tile))
    full = synth.make_latents(21, 60, 104, device="cuda").bfloat16()
    timed("vae decoder untiled 21x60x104", lambda: dec.decode(full))


if __name__ == "__main__":
    main()
