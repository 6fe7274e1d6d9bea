"""Sequence-sharded execution of joint_forward across the GPUs of one node (one process per GPU, RCCL over xGMI).

Why sequence sharding and not the head/column tensor parallelism a NVSwitch design would pick:
the path is one sample, so the only free axis is inside the forward.  Head/FFN-column TP needs an all-reduce of the
full [L, 5120] activation after every row-parallel GEMM (3 per DiT block, ~58 GB of all-reduce payload per forward),
and on MI355X's point-to-point xGMI mesh a ring all-reduce is bound by ONE link (~153 GB/s).  With 288 GB of HBM per
GPU the 36 GB of bf16 weights are simply replicated, every token-wise op (LayerNorm, all GEMMs, epilogues) runs on
L/N local rows with no communication, and the only exchange is an ALL-GATHER of the rotated K and V rows (and the
bicross q/k/v rows) in front of each attention: ~0.67 GB per DiT block instead of ~2 GB-equivalent of all-reduce,
no partial sums in reduced precision, no cross-GPU RMSNorm statistics (the full 5120-wide q/k RMSNorm stays local
to a token), and 12 bicross heads need not divide the GPU count.

Layout: DiT tokens are split into `world` contiguous row ranges (L = 32760 = 8 * 4095 at 480p); VGGT tokens are
split by whole frames (frame attention is per frame) -- 21 frames over 8 ranks = 3,3,3,3,3,2,2,2.
"""
from typing import List

import torch
import torch.distributed as dist


def split_counts(n: int, parts: int) -> List[int]:
    q, r = divmod(n, parts)
    return [q + (1 if i < r else 0) for i in range(parts)]


class SequenceShard:
    def __init__(self, rank: int, world: int, group=None):
        self.rank, self.world, self.group = rank, world, group
        self._grid = None

    # ---- per-grid bookkeeping -----------------------------------------------------------------------------------
    def _setup(self, F, hw, n_special):
        key = (F, hw, n_special)
        if self._grid == key:
            return
        self._grid = key
        L = F * hw
        P = n_special + hw
        self.dit_counts = split_counts(L, self.world)
        self.dit_start = sum(self.dit_counts[: self.rank])
        self.frame_counts = split_counts(F, self.world)
        self.first_frame = sum(self.frame_counts[: self.rank])
        self.my_frames = self.frame_counts[self.rank]
        if min(self.frame_counts) < 1:
            raise ValueError(f"{F} latent frames cannot be sharded over {self.world} ranks (need >= 1 frame per rank)")
        self.agg_counts = [c * P for c in self.frame_counts]
        self.P, self.hw = P, hw

    def localize_tables(self, tabs, F, hw, n_special):
        self._setup(F, hw, n_special)
        if "dit_local" in tabs and tabs.get("_shard_key") == (self.rank, self.world):
            return tabs
        s, n = self.dit_start, self.dit_counts[self.rank]
        a0 = self.first_frame * self.P
        tabs["dit_local"] = tabs["dit"][s:s + n].contiguous()
        tabs["bi_dit_local"] = tabs["bi_dit"][s:s + n].contiguous()
        tabs["bi_agg_local"] = tabs["bi_agg"][a0:a0 + self.my_frames * self.P].contiguous()
        tabs["_shard_key"] = (self.rank, self.world)
        return tabs

    # ---- data movement ------------------------------------------------------------------------------------------
    def take_dit_rows(self, t):
        return t[self.dit_start:self.dit_start + self.dit_counts[self.rank]].contiguous()

    def all_gather_rows(self, t, counts):
        """t: this rank's [counts[rank], C] rows (any strides) -> [sum(counts), C] with every rank's rows, in rank order."""
        assert t.shape[0] == counts[self.rank], (t.shape, counts, self.rank)
        t = t.contiguous()
        C = t.shape[1]
        mx = max(counts)
        if min(counts) == mx:
            out = torch.empty(mx * self.world, C, dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(out, t, group=self.group)
            return out
        pad = torch.zeros(mx, C, dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        buf = torch.empty(self.world, mx, C, dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(buf.view(self.world * mx, C), pad, group=self.group)
        return torch.cat([buf[r, : counts[r]] for r in range(self.world)], dim=0)

    def dit_rows_to_frames(self, ptok, hw):
        """Bridge (model_wan21.py:170-175): patch tokens are produced in the DiT row split but consumed per frame."""
        full = self.all_gather_rows(ptok, self.dit_counts)
        a = self.first_frame * hw
        return full[a:a + self.my_frames * hw].contiguous()

    def gather_frames(self, v):
        """[1, S_local, P, C] -> [1, S, P, C] (only for the layers the geometry heads read, last step only)."""
        _, s, P, C = v.shape
        rows = self.all_gather_rows(v.reshape(s * P, C), self.agg_counts)
        return rows.view(1, -1, P, C)


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT).  Returns
    (shard | None, rank, world, local_rank)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return None, 0, 1, local
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return SequenceShard(rank, world), rank, world, local
