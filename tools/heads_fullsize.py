"""Full-size geometry heads on one MI355X: 21 latent frames, 30x52 tokens, reference widths
(dim 2048, features 256, channels 256/512/1024/1024) -> pose_enc [1,81,9], depth / world_points [1,81,480,832,*].
Prints wall time per stage, peak memory, and the sanity properties (finite, confidences > 1, depth > 0)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fantasy_world_amd import config as fwc, synth                 # noqa: E402
from fantasy_world_amd.heads import GeometryHeads                  # noqa: E402
from fantasy_world_amd.hip_ops import HipOps                       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=21)
    ap.add_argument("--ph", type=int, default=30)
    ap.add_argument("--pw", type=int, default=52)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--gather", action="store_true", help="the older fw_im2col + fw_gemm_bf16 pair instead of the implicit-GEMM convolution")
    a = ap.parse_args()
    hc = fwc.HeadsConfig()
    ops = HipOps("cuda:0")
    t0 = time.time()
    W = synth.make_heads_weights(hc, device="cuda")
    print(f"weights: {sum(v.numel() for v in W.values())/1e6:.0f} M params in {time.time()-t0:.1f}s", flush=True)
    gh = GeometryHeads(hc, W.__getitem__, ops)
    gh.implicit_conv = not a.gather
    print("convolutions:", "gather + GEMM" if a.gather else "implicit GEMM", flush=True)
    del W
    torch.cuda.empty_cache()
    ol = {k: v[None] for k, v in synth.make_output_list(hc, a.frames, a.ph, a.pw, device="cuda").items()}
    torch.cuda.synchronize()
    for rep in range(a.reps):
        torch.cuda.reset_peak_memory_stats()
        t0 = time.time()
        pose = gh._camera(ol[max(ol)])
        torch.cuda.synchronize()
        t1 = time.time()
        d, dc = gh._dpt(gh.depth, ol, a.frames, a.ph, a.pw, 5)
        torch.cuda.synchronize()
        t2 = time.time()
        p, pc = gh._dpt(gh.point, ol, a.frames, a.ph, a.pw, 5)
        torch.cuda.synchronize()
        t3 = time.time()
        print(f"rep {rep}: camera {1e3*(t1-t0):.0f} ms, depth head {1e3*(t2-t1):.0f} ms, point head {1e3*(t3-t2):.0f} ms, "
              f"total {t3-t0:.2f} s, peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    print("shapes", tuple(pose.shape), tuple(d.shape), tuple(dc.shape), tuple(p.shape), tuple(pc.shape))
    for name, t in (("pose", pose), ("depth", d), ("depth_conf", dc), ("points", p), ("points_conf", pc)):
        assert torch.isfinite(t).all(), name
    assert (d > 0).all() and (dc >= 1).all() and (pc >= 1).all()
    print("finite, depth > 0, confidences >= 1: ok")


if __name__ == "__main__":
    main()
