#!/usr/bin/env python
"""Per-rank compute time of a sharded forward, measured on ONE GPU: the collectives are replaced by local stand-ins that return
tensors of the right shape, so every kernel runs at exactly the shapes a rank of an n-way group sees.
  --sp 1,2,4   sequence shard (L/n rows for GEMMs and norms, L rows x H/n heads for attention; parallel.py)
  --tp 2,4,8   north_star's head / FFN-column tensor parallelism (all L rows, H/n heads, 1/n of the FFN columns, replicated norms
               and fw_residual_add epilogues; tensor_parallel.py), all-reduce = no-op
compute-only scaling bound = t(1) / t(n); communication is NOT included (docs/multi_gpu.md has the byte counts of both)."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fantasy_world_amd import config as fwc, synth
from fantasy_world_amd.engine import FusionEngine
from fantasy_world_amd.hip_ops import HipOps
from fantasy_world_amd.parallel import Ready, SequenceShard


class LocalShard(SequenceShard):
    """SequenceShard whose collectives complete locally with tensors of the right shape (own data tiled)."""

    def _tile(self, t, counts):
        total = sum(counts)
        reps = (total + t.shape[0] - 1) // t.shape[0]
        return t.repeat(reps, *([1] * (t.dim() - 1)))[:total].contiguous()

    def all_gather_rows_async(self, t, counts):
        return Ready(self._tile(t.contiguous(), counts))

    def rows_to_heads_async(self, t, parts, counts, cols=None):
        rows, width = t.shape
        c = width // (parts * self.world)
        mine = t.reshape(rows, parts, self.world, c)[:, :, self.rank, :]
        if cols is not None:
            mine = mine[:, :, cols[0]:cols[1]]
        return Ready(self._tile(mine.contiguous(), counts))

    def heads_to_rows_async(self, o, counts):
        rows = counts[self.rank]
        return Ready(o[:rows].repeat(1, self.world).contiguous())


def local_tensor_shard(n):
    from fantasy_world_amd.tensor_parallel import TensorShard

    class LocalTensorShard(TensorShard):
        def all_reduce_async(self, t, kind="all_reduce"):
            return Ready(t)

        def all_gather_rows_async(self, t, counts):
            total = sum(counts)
            reps = (total + t.shape[0] - 1) // t.shape[0]
            return Ready(t.repeat(reps, 1)[:total].contiguous())

    return LocalTensorShard(0, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sp", default="1,2,4,8")
    ap.add_argument("--tp", default="", help="tensor-parallel degrees to emulate after the sequence-shard ones, e.g. 2,4,8")
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--host-cost", action="store_true",
                    help="also run the SAME full-depth engine (same launch count, same Python path) on a tiny latent grid, where "
                         "the GPU finishes every kernel in microseconds: that forward's wall time IS the host's enqueue cost "
                         "(Python + ctypes + torch.empty per op) -- the number that decides whether a HIP graph would pay")
    args = ap.parse_args()
    dev = "cuda:0"
    ops = HipOps(dev)
    cfg = fwc.wan21_14b()
    spec = synth.weight_spec(cfg)
    F, H2, W2 = 21, 60, 104
    ins = synth.make_inputs(cfg, F, H2, W2, seed=1, device=dev, dtype=torch.bfloat16)
    cond = dict(clip_feature=ins["clip_feature"], y=ins["y"], plucker_fea=ins["plucker_fea"],
                plucker_context_lens=ins["plucker_context_lens"])
    t = torch.tensor([500.0], device=dev, dtype=torch.bfloat16)
    base = None
    plan = [("sp", int(v)) for v in args.sp.split(",") if v] + [("tp", int(v)) for v in args.tp.split(",") if v]
    for kind, n in plan:
        get = lambda nm: synth.make_param(nm, spec[nm][0], spec[nm][1], device=dev)
        if kind == "tp":
            from fantasy_world_amd.tensor_parallel import TPFusionEngine
            eng = TPFusionEngine(cfg, get, ops, local_tensor_shard(n))
        else:
            eng = FusionEngine(cfg, get, ops, shard=None if n == 1 else LocalShard(0, n))
        eng.joint_forward(ins["x"], t, ins["context"], **cond)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(args.iters):
            eng.joint_forward(ins["x"], t, ins["context"], **cond)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / args.iters
        base = base or dt
        print(f"{kind}={n}: one forward (rank 0 shapes) {dt*1e3:8.1f} ms   compute-only speed-up vs sp=1: {base/dt:5.2f}x   "
              f"(x{2 if True else 1} CFG groups -> {2*n} GPUs: step = {dt*1e3:.0f} ms + comm)", flush=True)
        if args.host_cost:
            tiny = synth.make_inputs(cfg, 8, 8, 8, seed=1, device=dev, dtype=torch.bfloat16)      # 8 latent frames: >= 1 per rank
            tc = dict(clip_feature=tiny["clip_feature"], y=tiny["y"], plucker_fea=tiny["plucker_fea"],
                      plucker_context_lens=tiny["plucker_context_lens"])
            for _ in range(2):
                eng.joint_forward(tiny["x"], t, tiny["context"], **tc)
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(5):
                eng.joint_forward(tiny["x"], t, tiny["context"], **tc)
            torch.cuda.synchronize()
            th = (time.time() - t0) / 5
            print(f"        host enqueue cost of one forward at {kind}={n} (tiny grid, launch-bound): {th*1e3:6.1f} ms = "
                  f"{100*th/dt:4.1f} % of the full-size forward's {dt*1e3:.0f} ms", flush=True)
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
