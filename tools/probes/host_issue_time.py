"""How long the host needs to ENQUEUE one full-size joint_forward (Python + ctypes + torch.empty per op) versus how long the
GPU needs to run it: the margin that decides whether a HIP graph would buy anything, also at 4-way sequence sharding where the
kernels are 4x shorter but the op count is the same."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fantasy_world_amd import config as fwc, synth                 # noqa: E402
from fantasy_world_amd.engine import FusionEngine                  # noqa: E402
from fantasy_world_amd.hip_ops import HipOps                       # noqa: E402

dev = "cuda:0"
ops = HipOps(dev)
cfg = fwc.wan21_14b()
spec = synth.weight_spec(cfg)
eng = FusionEngine(cfg, lambda n: synth.make_param(n, spec[n][0], spec[n][1], device=dev), ops)
ins = synth.make_inputs(cfg, 21, 60, 104, seed=1, device=dev, dtype=torch.bfloat16)
cond = dict(clip_feature=ins["clip_feature"], y=ins["y"], plucker_fea=ins["plucker_fea"], plucker_context_lens=ins["plucker_context_lens"])
t = torch.tensor([500.0], device=dev, dtype=torch.bfloat16)
for _ in range(2):
    eng.joint_forward(ins["x"], t, ins["context"], **cond)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.time()
    eng.joint_forward(ins["x"], t, ins["context"], **cond)
    t1 = time.time()
    torch.cuda.synchronize()
    t2 = time.time()
    print(f"rep {rep}: host enqueue {1e3*(t1-t0):.0f} ms, GPU done after {1e3*(t2-t0):.0f} ms", flush=True)
