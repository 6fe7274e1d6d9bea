"""Multi-GPU execution of the denoising step on one node (one process per GPU, RCCL over xGMI).

Two levels, chosen from what the path offers and what the xGMI mesh (7 point-to-point links x ~153 GB/s per GPU) is good at:

1. CFG parallelism (outer level, 2 groups).  A denoise step is two INDEPENDENT joint_forward calls (positive / negative
   prompt, FantasyWorld/fusion/model_wan21.py:295-319) followed by a 4 MB combine.  With an even world size the ranks are split
   into two groups, each group runs one of the two forwards, and the only exchange is one all-gather of noise_pred per step.
   At 2 GPUs the forward needs no collective at all.

2. Sequence sharding inside a group (world/2 ranks; world ranks when the world size is odd).  The 36 GB of bf16 weights
   are replicated (288 GB HBM per GPU); LayerNorm, every GEMM and every fused epilogue run on L/n local rows with no
   communication -- the GEMMs keep M >= 8190 rows at 8 GPUs, which matters because the 256x256-tile kernels quantise badly
   below that (M = 4095, N = 5120 is 320 tiles on 256 CUs = 2 rounds).  Only attention needs other ranks' tokens:
     * DiT self-attention and VGGT global attention: head exchange (all-to-all): the rotated q|k|v rows [L/n, 3*H*hd] go out
       as n column blocks of H/n heads and come back as [L, 3*(H/n)*hd]; attention runs over the FULL sequence for H/n heads;
       the output returns by the inverse all-to-all.  Per DiT block each GPU sends (n-1)/n of 126 MB + 42 MB instead of
       receiving (n-1)/n of the 671 MB k|v all-gather (4x fewer bytes), and still nothing is reduced in low precision.
     * bicross attention (12 heads): head exchange as well when the group size divides 12 (2, 3, 4, 6 ranks: the 2 x 4 layout of 8
       GPUs; round 3) -- q|v1 and k|v2 out, both directions over the full sequences for 12/n heads, o1 / o2 back; otherwise, and for
       any head count that does not divide: all-gather of the rotated rows (q|v1: 151 MB, k|v2: 151 MB per IRG block).
   Head/FFN-column tensor parallelism (the NVSwitch habit) would instead all-reduce the full [L,5120] activation three times
   per DiT block (1.76 GB received per GPU per block, partial sums rounded to bf16, cross-GPU statistics for the full-width
   q/k RMSNorm) -- see docs/multi_gpu.md for the byte table.

Layout: DiT tokens are split into contiguous row ranges (L = 32760 = 4 * 8190), VGGT tokens by whole frames (frame attention
is per frame; 21 frames over 4 ranks = 6,5,5,5).
"""
import os
import time
from typing import List

# The host driver of the MI355X boxes only supports dmabuf IPC: without this RCCL (and any CUDA-tensor sharing across processes) fails
# with `hipIpcGetMemHandle: invalid argument` at the first collective.  Set before the HSA runtime starts, however the ranks were
# launched (the driver's own `python -m torch.distributed.run ... bench.py`, a user's torchrun around the reference script, ...).
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist


def split_counts(n: int, parts: int) -> List[int]:
    q, r = divmod(n, parts)
    return [q + (1 if i < r else 0) for i in range(parts)]


class CommStats:
    """Observability of the exchanges (bench.py `comm` block).  Per collective: the bytes this rank puts on the wire, and --
    on a HIP device -- three events on the COMPUTE stream: at issue, just before wait() and just after it.  The compute
    stream only blocks inside wait(), so  exposed = after - before  is the time the exchange was NOT hidden behind compute, and
    window = after - issue  bounds the exchange's duration from above (it includes the compute enqueued in between).  On CPU
    tensors (gloo tests, dry runs) the host clock around wait() stands in for the events."""

    def __init__(self):
        self.records = []           # [kind, bytes, ev_issue, ev_before, ev_after] / host seconds on CPU

    def issue(self, kind, nbytes, tensor):
        rec = [kind, int(nbytes), None, None, None]
        if tensor.is_cuda:
            rec[2] = torch.cuda.Event(enable_timing=True)
            rec[2].record()
        self.records.append(rec)
        return rec

    def summary(self, steps):
        """Call after a device synchronize.  Per step and per GPU: bytes sent, exposed ms, window ms, number of collectives."""
        by_kind = {}
        tot_b = tot_e = tot_w = 0.0
        for kind, nbytes, e0, e1, e2 in self.records:
            if e2 is None:
                continue
            if isinstance(e1, float):
                exposed, window = 1e3 * (e2 - e1), 1e3 * (e2 - e1)
            else:
                exposed, window = e1.elapsed_time(e2), e0.elapsed_time(e2)
            k = by_kind.setdefault(kind, dict(n=0, bytes=0.0, exposed_ms=0.0, window_ms=0.0))
            k["n"] += 1
            k["bytes"] += nbytes
            k["exposed_ms"] += exposed
            k["window_ms"] += window
            tot_b, tot_e, tot_w = tot_b + nbytes, tot_e + exposed, tot_w + window
        steps = max(steps, 1)
        return {"bytes_sent_per_gpu_per_step": tot_b / steps, "exposed_ms_per_step": tot_e / steps,
                "issue_to_done_ms_per_step": tot_w / steps, "collectives_per_step": len(self.records) / steps,
                "by_kind": {k: {a: (b / steps) for a, b in v.items()} for k, v in by_kind.items()}}


STATS = None          # set by enable_comm_stats(); None = no instrumentation (the default)


def enable_comm_stats():
    global STATS
    STATS = CommStats()
    return STATS


def disable_comm_stats():
    global STATS
    STATS = None


class Pending:
    """A collective in flight.  torch.distributed runs it on the backend's own stream (RCCL: a side HIP stream that first
    waits for the producer kernels already enqueued on the compute stream); wait() makes the COMPUTE stream wait for it (no
    host sync) and returns the result.  Everything enqueued between the issue and wait() overlaps the exchange."""

    __slots__ = ("_work", "_finish", "_rec")

    def __init__(self, work, finish, kind=None, nbytes=0, tensor=None):
        self._work, self._finish = work, finish
        self._rec = STATS.issue(kind, nbytes, tensor) if (STATS is not None and work is not None and kind) else None

    def wait(self):
        if self._work is not None:
            rec = self._rec
            if rec is not None:
                if rec[2] is not None:
                    rec[3] = torch.cuda.Event(enable_timing=True)
                    rec[3].record()
                else:
                    import time
                    rec[3] = time.perf_counter()
            self._work.wait()
            self._work = None
            if rec is not None:
                if rec[2] is not None:
                    rec[4] = torch.cuda.Event(enable_timing=True)
                    rec[4].record()
                else:
                    import time
                    rec[4] = time.perf_counter()
        return self._finish()


class Ready(Pending):
    """Same interface for a value that needs no communication (single-rank group)."""

    def __init__(self, value):
        super().__init__(None, lambda: value)


class SequenceShard:
    def __init__(self, rank: int, world: int, group=None, probe_group=None):
        self.rank, self.world, self.group = rank, world, group
        self.probe_group = probe_group        # a second communicator over the same ranks, used ONLY by probe_grouped_exchange
        self._grid = None
        self.exchange_probe = None            # {"requested", "ran", "ok", "seconds"} once negotiated (bench.py `comm` block)

    def negotiate_exchange_groups(self, requested, device, timeout_s=None):
        """How many head groups a DiT self-attention exchange may be cut into (FusionEngine.exchange_groups).  More than one group
        means a SECOND all-to-all is issued while the first is still in flight (and the inverse exchange of group g while group
        g+1 travels).  RCCL runs a communicator's collectives in issue order on its own stream, where that is fine -- but it has
        never run on more than one rank here, and gloo's device staging stalled on exactly this pattern
        (profiles/r03/dryrun_ranks_small.txt).  So the pattern is tried once, on small tensors, on the PROBE communicator: if it does
        not complete within `timeout_s` ($FW_SP_PROBE_TIMEOUT_S, default 20) on every rank, the engine falls back to one exchange
        per attention (every rank takes the same decision: MIN over the group) and the probe communicator is never used again (the
        stuck collectives stay on it, away from the communicator the forward uses).  The outcome is kept in `exchange_probe`."""
        if self.exchange_probe is not None and self.exchange_probe["requested"] == requested:
            return self.exchange_probe["ran"]
        ran, ok, secs = requested, True, 0.0
        if requested > 1 and self.world > 1 and self.probe_group is None:
            # no side communicator to try the pattern on: never probe on the forward's own communicator (a stall would strand its
            # collectives exactly where the forward needs to run) -- one exchange per attention, recorded as not probed
            self.exchange_probe = {"requested": int(requested), "ran": 1, "ok": True, "seconds": 0.0, "probed": False,
                                   "note": "no probe communicator: grouped exchange not used"}
            return 1
        if requested > 1 and self.world > 1:
            if timeout_s is None:
                timeout_s = float(os.environ.get("FW_SP_PROBE_TIMEOUT_S", "20"))
            t0 = time.monotonic()
            ok = self._probe_grouped_exchange(device, timeout_s)
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)      # plain blocking collective on the forward's communicator
            ok = bool(int(flag.item()))
            secs = time.monotonic() - t0
            if not ok:
                ran = 1
                # a probe that really stalled leaves collectives -- device kernels under RCCL -- on the probe communicator: every later
                # device-wide synchronize (bench.py's barrier) and the process-group watchdog would wait on them.  Abort that
                # communicator NOW (ncclCommAbort ends its kernels); if it cannot be aborted the run is refused instead of hanging.
                self._abandon_probe_group()
        self.exchange_probe = {"requested": int(requested), "ran": int(ran), "ok": bool(ok), "seconds": round(secs, 3)}
        return ran

    def _abandon_probe_group(self):
        """After a failed probe: end whatever is still in flight on the probe communicator so that nothing on this device waits for
        it.  gloo (CPU tests, the several-ranks-on-one-GPU debugging aid): nothing runs on the device, nothing to do.  RCCL: abort the
        communicator; a backend without abort() -> RuntimeError (a clear refusal beats a 900 s hang)."""
        g, self.probe_group = self.probe_group, None
        if g is None:
            return "no probe communicator"
        name = dist.get_backend(g)
        if name != "nccl":
            return f"{name}: nothing on the device"
        try:
            be = g._get_backend(torch.device("cuda"))
            be.abort()
        except Exception as e:
            raise RuntimeError("the grouped head-exchange probe did not complete on this rank set and its RCCL communicator could not "
                               f"be aborted ({e!r}): refusing to continue -- set FW_SP_EXCHANGE_GROUPS=1 to skip the probe") from e
        self._probe_keepalive = None
        return "nccl: probe communicator aborted"

    def _probe_grouped_exchange(self, device, timeout_s, rounds=4, rows=512, c=256):
        """The engine's own pattern (FusionEngine._dit_attn_begin / _dit_attn_mid), `rounds` blocks back to back on MB-sized bf16
        messages: both q|k|v group exchanges in flight, wait for group 0, start its inverse (output) exchange while group 1 still
        travels, wait for group 1, start its inverse exchange, wait for both outputs.  Returns False as soon as one wait does not
        complete by the deadline (host polling of Work.is_completed(), no blocking wait), or if the data that came back is wrong."""
        if os.environ.get("FW_SP_PROBE_FORCE_FAIL") == str(self.rank):       # test hook: this rank reports a stall
            return False
        n = self.world
        pr = SequenceShard(self.rank, n, self.probe_group if self.probe_group is not None else self.group)
        counts = [rows] * n
        base = (torch.arange(rows * 3 * n * 2 * c, dtype=torch.float32, device=device) % 251).view(rows, 3 * n * 2 * c)
        qkv = (base + self.rank).to(torch.bfloat16)
        deadline = time.monotonic() + timeout_s
        self._probe_keepalive = keep = [qkv]                # a stuck collective must not be destroyed under the backend

        def done(p):
            while not p._work.is_completed():
                if time.monotonic() > deadline:
                    return False
                time.sleep(0.002)
            return True
        for _ in range(rounds):
            pend = [pr.rows_to_heads_async(qkv, 3, counts, (0, c)), pr.rows_to_heads_async(qkv, 3, counts, (c, 2 * c))]
            keep += pend
            back = []
            for p_ in pend:
                if not done(p_):
                    return False
                got = p_.wait()                               # [rows * n, 3, c]: every rank's rows for my heads
                back.append(pr.heads_to_rows_async(got[:, 0].contiguous(), counts))
                keep.append(back[-1])
            outs = []
            for b_ in back:
                if not done(b_):
                    return False
                outs.append(b_.wait())                        # [rows, n * c]: my rows, every rank's head block of q
            # exchange followed by its inverse is the identity on q's columns of this group
            q = qkv.view(rows, 3, n, 2 * c)[:, 0]              # [rows, n, 2c]
            for g, o in enumerate(outs):
                if not torch.equal(o.view(rows, n, c), q[:, :, g * c:(g + 1) * c]):
                    return False
        self._probe_keepalive = None
        return True

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

    def all_gather_rows_async(self, t, counts) -> Pending:
        """t: this rank's [counts[rank], C] rows (any strides) -> Pending of [sum(counts), C] (every rank's rows, rank order)."""
        assert t.shape[0] == counts[self.rank], (t.shape, counts, self.rank)
        t = t.contiguous()
        C = t.shape[1]
        mx = max(counts)
        if min(counts) == mx:
            out = torch.empty(mx * self.world, C, dtype=t.dtype, device=t.device)
            work = dist.all_gather_into_tensor(out, t, group=self.group, async_op=True)
            return Pending(work, lambda: out, "all_gather_rows", self._sent(t.numel() * t.element_size()), t)
        pad = torch.zeros(mx, C, dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        buf = torch.empty(self.world, mx, C, dtype=t.dtype, device=t.device)
        work = dist.all_gather_into_tensor(buf.view(self.world * mx, C), pad, group=self.group, async_op=True)
        return Pending(work, lambda: torch.cat([buf[r, : counts[r]] for r in range(self.world)], dim=0),
                       "all_gather_rows", self._sent(pad.numel() * pad.element_size()), pad)

    def all_gather_rows(self, t, counts):
        return self.all_gather_rows_async(t, counts).wait()

    def _sent(self, my_bytes):
        """Bytes this rank sends in an all-gather of `my_bytes` per rank: its block goes to each of the other world-1 ranks
        (direct exchange over the xGMI mesh; a ring would forward the same total)."""
        return my_bytes * (self.world - 1)

    def heads_divisible(self, heads):
        return heads % self.world == 0

    def rows_to_heads_async(self, t, parts, counts, cols=None) -> Pending:
        """Head exchange, forward direction.  t: this rank's rows [counts[rank], parts*H*hd] laid out as `parts` blocks of
        H*hd columns (q|k|v).  Pending of [sum(counts), parts, (H/world)*hd]: every rank's rows (rank order = token order)
        for this rank's H/world heads.  One all_to_all_single with uneven row splits.
        cols = (a, b): only columns [a, b) of every rank's (H/world)*hd block -- a group of local heads -- so that the
        exchange of the next group can run behind the attention of this one."""
        rows, width = t.shape
        assert rows == counts[self.rank] and width % (parts * self.world) == 0, (t.shape, parts, counts)
        c = width // (parts * self.world)                       # (H/world)*hd
        src = t.reshape(rows, parts, self.world, c)
        if cols is not None:
            src = src[:, :, :, cols[0]:cols[1]]
            c = cols[1] - cols[0]
        send = src.permute(2, 0, 1, 3).contiguous()                                         # [world, rows, parts, c]
        out = torch.empty(sum(counts), parts, c, dtype=t.dtype, device=t.device)
        work = dist.all_to_all_single(out.view(sum(counts), parts * c), send.view(self.world * rows, parts * c),
                                      output_split_sizes=list(counts), input_split_sizes=[rows] * self.world,
                                      group=self.group, async_op=True)
        # (world-1)/world of the send buffer leaves this GPU (its own block stays)
        return Pending(work, lambda: out, "all_to_all_qkv", send.numel() * send.element_size() * (self.world - 1) // self.world, send)

    def rows_to_heads(self, t, parts, counts):
        return self.rows_to_heads_async(t, parts, counts).wait()

    def heads_to_rows_async(self, o, counts) -> Pending:
        """Inverse exchange for the attention output.  o: [sum(counts), (H/world)*hd] (all rows, my heads) ->
        Pending of [counts[rank], H*hd] (my rows, all heads)."""
        total, c = o.shape
        assert total == sum(counts), (o.shape, counts)
        rows = counts[self.rank]
        recv = torch.empty(self.world * rows, c, dtype=o.dtype, device=o.device)
        src = o.contiguous()
        work = dist.all_to_all_single(recv, src, output_split_sizes=[rows] * self.world,
                                      input_split_sizes=list(counts), group=self.group, async_op=True)
        return Pending(work, lambda: recv.view(self.world, rows, c).permute(1, 0, 2).reshape(rows, self.world * c),
                       "all_to_all_out", (total - rows) * c * src.element_size(), src)

    def heads_to_rows(self, o, counts):
        return self.heads_to_rows_async(o, counts).wait()

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


class Topology:
    """Who this process is: world rank, CFG group (0 = positive prompt, 1 = negative) and, inside the group, either the sequence
    shard (`shard`, mode "sp": head all-to-all, this file) or the tensor-parallel shard (`tp`, mode "tp": head / FFN-column split
    with all-reduce, tensor_parallel.py -- north_star's partition)."""

    def __init__(self, rank=0, world=1, local=0, cfg_groups=1, cfg_rank=0, shard=None, tp=None):
        self.rank, self.world, self.local = rank, world, local
        self.cfg_groups, self.cfg_rank, self.shard, self.tp = cfg_groups, cfg_rank, shard, tp

    @property
    def mode(self):
        return "tp" if self.tp is not None else "sp"

    @property
    def sp_world(self):
        return 1 if self.shard is None else self.shard.world

    @property
    def group_world(self):
        """Ranks that share ONE forward (sequence-shard or tensor-parallel degree)."""
        return self.tp.world if self.tp is not None else self.sp_world

    def describe(self):
        if self.world == 1:
            return "single GPU"
        parts = []
        if self.cfg_groups == 2:
            parts.append("CFG-parallel x2")
        if self.sp_world > 1:
            parts.append(f"sequence-sharded x{self.sp_world} (head all-to-all + K/V all-gather over RCCL)")
        if self.tp is not None and self.tp.world > 1:
            parts.append(f"tensor-parallel x{self.tp.world} (attention heads / FFN columns, all-reduce over RCCL)")
        return " x ".join(parts)

    def gather_cfg(self, out):
        """[pos, neg] noise predictions from the two CFG groups (one all-gather over the world group per step)."""
        bufs = [torch.empty_like(out) for _ in range(self.world)]
        src = out.contiguous()
        work = dist.all_gather(bufs, src, async_op=True)
        Pending(work, lambda: None, "all_gather_cfg", src.numel() * src.element_size() * (self.world - 1), src).wait()
        return bufs[0], bufs[self.world // 2]


def make_topology(rank, world, local=0, cfg_parallel=True, mode="sp", reduce_dtype=None):
    """Process groups for `world` ranks: two CFG groups of world/2 ranks when world is even (and cfg_parallel), each
    sharded internally; otherwise one group.  mode "sp" = sequence shard with head exchange (SequenceShard), "tp" = head /
    FFN-column tensor parallelism with all-reduce (tensor_parallel.TensorShard).  Every rank must call this (new_group is
    collective)."""
    if mode not in ("sp", "tp"):
        raise ValueError(f"mode must be 'sp' or 'tp', got {mode!r}")
    if world == 1:
        return Topology()

    def inner(r, n, group):
        if n == 1:
            return {}
        if mode == "tp":
            from .tensor_parallel import TensorShard
            kw = {} if reduce_dtype is None else dict(reduce_dtype=reduce_dtype)
            return dict(tp=TensorShard(r, n, group[0], **kw))
        return dict(shard=SequenceShard(r, n, group[0], probe_group=group[1]))

    def groups_for(ranks):
        # (the forward's communicator, the probe communicator of SequenceShard.negotiate_exchange_groups); new_group is collective
        # over the WORLD: every rank creates every group, in the same order
        main = dist.new_group(ranks)
        probe = dist.new_group(ranks) if (mode == "sp" and len(ranks) > 1) else None
        return main, probe

    if cfg_parallel and world % 2 == 0:
        n = world // 2
        groups = [groups_for(list(range(g * n, (g + 1) * n))) for g in range(2)]
        cfg_rank, sp_rank = rank // n, rank % n
        return Topology(rank, world, local, 2, cfg_rank, **inner(sp_rank, n, groups[cfg_rank]))
    return Topology(rank, world, local, 1, 0, **inner(rank, world, groups_for(list(range(world)))))


def reduce_dtype_from_env():
    """FW_TP_REDUCE_DTYPE: dtype of the all-reduced partial sums of the tensor-parallel partition.  Default fp32, decided from the
    parity data (tests/golden/parity_bounds_gpu.json `tp/world4/*`): with fp32 partial sums the sharded forward sits at the
    unsharded one's distance from the fp32 golden (2.68e-3), with bf16 partial sums at 3.2e-3 -- +20 % error for half the bytes.
    What the bytes cost is measured by bench.py at N > 1 (`comm.microbench`: the same [rows, 5120] all-reduce in both dtypes),
    so the first hardware run records the price of this default next to it; bf16 stays one environment variable away."""
    reduce_dtype = {"fp32": torch.float32, "bf16": torch.bfloat16}.get(os.environ.get("FW_TP_REDUCE_DTYPE", "fp32"))
    if reduce_dtype is None:
        raise ValueError("FW_TP_REDUCE_DTYPE must be fp32 or bf16")
    return reduce_dtype


def alt_topology(topo):
    """The OTHER partition over the same ranks (bench.py's `alt` block at N > 1: one hardware opportunity answers both partition
    questions): tensor-parallel groups when `topo` is sequence-sharded and vice versa, same CFG split.  New
    process groups are created (collective over the world: every rank must call this).  None when the ranks that share one
    forward are a single rank (world 2 = two CFG groups of one GPU: both partitions are the unsharded forward)."""
    if topo.world == 1 or topo.group_world == 1:
        return None
    mode = "sp" if topo.tp is not None else "tp"
    return make_topology(topo.rank, topo.world, topo.local, cfg_parallel=topo.cfg_groups == 2, mode=mode,
                         reduce_dtype=reduce_dtype_from_env())


def init_topology(backend=None, cfg_parallel=True, mode=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT) -> Topology.
    mode: "sp" | "tp"; default from $FW_PARALLEL, else "sp"."""
    mode = mode or os.environ.get("FW_PARALLEL", "sp")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return Topology(local=local)
    reduce_dtype = reduce_dtype_from_env()
    if not dist.is_initialized():
        if backend is None:
            # "nccl" IS RCCL on ROCm.  FW_DIST_BACKEND=gloo: debugging aid for the CFG-parallel path only (several ranks sharing
            # one GPU, where RCCL refuses); gloo's all-to-all on device tensors does not make progress, so no sequence shard with it
            backend = os.environ.get("FW_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return make_topology(rank, world, local, cfg_parallel, mode=mode, reduce_dtype=reduce_dtype)


def make_engine(cfg, get, ops, topo=None, **kw):
    """The engine for this rank's place in `topo`: FusionEngine (single GPU / sequence shard) or TPFusionEngine."""
    if topo is not None and topo.tp is not None:
        from .tensor_parallel import TPFusionEngine
        return TPFusionEngine(cfg, get, ops, topo.tp, **kw)
    from .engine import FusionEngine
    eng = FusionEngine(cfg, get, ops, shard=None if topo is None else topo.shard, **kw)
    if eng.shard is not None and eng.shard.world > 1:
        # the grouped q|k|v exchange is tried once on a side communicator; a rank set on which it does not complete runs one
        # exchange per attention instead (SequenceShard.negotiate_exchange_groups)
        dev = getattr(ops, "device", None) or "cpu"
        eng.exchange_groups = eng.shard.negotiate_exchange_groups(eng.exchange_groups, dev)
    return eng


def init_from_env(backend=None):
    """Back-compat: (shard | None, rank, world, local_rank) with ONE sequence-sharded group (no CFG split)."""
    topo = init_topology(backend, cfg_parallel=False)
    return topo.shard, topo.rank, topo.world, topo.local


GOLDEN_TOL = {"bf16": 8e-3, "fp8": 8e-2}     # rel-L2 of noise_pred against the fp32 reference golden; fp8: e4m3's own price (measured
                                             # 4.7e-2 on one GPU, tests/golden/parity_bounds_gpu.json `fp8/wan21_cfg1...`, fp8 attention included)


def golden_self_check(topo, ops, case="wan21_cfg1_l2_f9_64x64", tol=None, golden_dir=None, precision="bf16", fp8_attention=False):
    """First-run safety of a multi-rank measurement (bench.py at N > 1): the small golden case -- the REAL reference's fp32
    joint_forward on BASELINE configs[0] (2-block model, latents [1,16,9,64,64]; tests/golden/, generated by oracle/make_golden.py
    from the imported reference) -- through THIS rank's place in `topo` (its sequence shard / tensor-parallel shard, every exchange
    the measured forward makes).  Returns {"case", "tol", "rel_l2" (MAX over ranks), "ok"}; a rank set that is off the golden must
    not print a throughput.  9 latent frames shard over up to 8 ranks.
    precision / fp8_attention: the engine options of the run being guarded (BASELINE config 5: fp8 linears + fp8 attention under
    either partition); the fp8 engines are held to e4m3's own distance from the fp32 golden (GOLDEN_TOL)."""
    tol = GOLDEN_TOL[precision] if tol is None else tol
    import torch.distributed as dist
    from . import config as fwc, synth
    # the default case ships INSIDE the package (fantasy_world_amd/golden/: meta + noise_pred of the tests/golden fixture, byte-checked by
    # tests/test_multi_gpu_first_run_cpu.py) so that an installed package without the tests/ tree keeps its first-run check; any other
    # case is looked up in the repository's tests/golden/.  A missing fixture RAISES: the check must never vanish silently.
    here = os.path.dirname(os.path.abspath(__file__))
    cands = [golden_dir] if golden_dir else [os.path.join(here, "golden"), os.path.join(os.path.dirname(here), "tests", "golden")]
    path = next((os.path.join(d, case + ".pt") for d in cands if os.path.isfile(os.path.join(d, case + ".pt"))), None)
    if path is None:
        raise FileNotFoundError(f"golden_self_check: fixture {case}.pt not found in {cands}; the multi-rank first-run check cannot run "
                                "(FW_BENCH_GOLDEN_CHECK=0 skips it deliberately)")
    g = torch.load(path, map_location="cpu", weights_only=False)
    meta = g["meta"]
    cfg = (fwc.plumbing22 if meta.get("flavour") == "wan22" else fwc.plumbing)(**meta["cfg"])
    f, h2, w2 = meta["grid"]
    W = synth.LazyWeights(synth.weight_spec(cfg), seed=meta["seed_weights"])          # CPU generators: the golden's own weights
    dev = getattr(ops, "device", None) or "cpu"
    eng = make_engine(cfg, W.__getitem__, ops, topo, precision=precision, **({"fp8_attention": True} if fp8_attention else {}))
    ins = synth.make_inputs(cfg, f, h2, w2, seed=meta["seed_inputs"], timestep=meta["timestep"], text_len=meta["text_len"], device=dev)
    kw = dict(y=ins["y"])
    if cfg.control_adapter:
        kw["control_camera_latents_input"] = ins["control_camera_latents_input"]
    else:
        kw.update(clip_feature=ins["clip_feature"], plucker_fea=ins["plucker_fea"], plucker_context_lens=ins["plucker_context_lens"])
    out, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], uncond=meta["uncond"], **kw)
    want = g["noise_pred"].to(out.device).double()
    err = ((out.double() - want).norm() / want.norm()).reshape(1).to(torch.float64)
    if not torch.isfinite(err).all():
        err = torch.full_like(err, float("inf"))
    if topo is not None and topo.world > 1:
        dist.all_reduce(err, op=dist.ReduceOp.MAX)
    e = float(err.item())
    del eng
    return {"case": case, "tol": tol, "rel_l2_max_over_ranks": e, "ok": bool(e < tol), "precision": precision,
            "fp8_attention": bool(fp8_attention)}


def comm_microbench(topo, device, rows, width=5120, iters=3):
    """What the bytes cost on THIS machine's links, measured once beside a multi-rank bench line (`comm.microbench`): the
    [rows, width] all-reduce of the tensor-parallel partition in bf16 and in fp32 (the FW_TP_REDUCE_DTYPE decision), and the q|k|v
    head all-to-all of the sequence shard, on the rank group that shares one forward.  ms per call, MAX over the group's ranks."""
    import torch.distributed as dist
    group = topo.tp.group if topo.tp is not None else (topo.shard.group if topo.shard is not None else None)
    n = topo.group_world
    if n <= 1:
        return None
    cuda = torch.device(device).type == "cuda"

    def timed(fn):
        fn()
        if cuda:
            torch.cuda.synchronize()
        dist.barrier(group=group)
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        if cuda:
            torch.cuda.synchronize()
        dt = torch.tensor([1e3 * (time.perf_counter() - t0) / iters], dtype=torch.float64, device=device)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX, group=group)
        return float(dt.item())
    out = {"rows": int(rows), "width": int(width), "ranks": int(n)}
    # everything that can fail on ONE rank (allocation) happens before the first collective, and whether to run is decided
    # collectively (MIN of an ok flag): a rank that raised would otherwise leave the others blocked in the all-reduce
    bufs, err = {}, None
    try:
        if os.environ.get("FW_COMM_MICROBENCH_FORCE_FAIL") == str(topo.rank):       # test hook
            raise MemoryError("forced")
        for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
            bufs[name] = torch.ones(rows, width, dtype=dt, device=device)
        r = rows // n
        bufs["src"] = torch.ones(n * r, 3 * width // n, dtype=torch.bfloat16, device=device)
        bufs["dst"] = torch.empty_like(bufs["src"])
    except Exception as e:
        err = repr(e)[:200]
    flag = torch.tensor([0 if err else 1], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 0:
        return {"error": err or "another rank of the group could not allocate the buffers", "rows": int(rows), "ranks": int(n)}
    for name in ("bf16", "fp32"):
        t = bufs[name]
        out[f"all_reduce_{name}_ms"] = timed(lambda: dist.all_reduce(t, group=group))
        out[f"all_reduce_{name}_bytes"] = t.numel() * t.element_size()
    src, dst = bufs["src"], bufs["dst"]
    out["all_to_all_qkv_bf16_ms"] = timed(lambda: dist.all_to_all_single(dst, src, group=group))
    out["all_to_all_qkv_bf16_bytes"] = src.numel() * 2
    return out
