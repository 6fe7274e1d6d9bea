"""north_star's partition on CPU over gloo: head / FFN-column tensor parallelism with all-reduce (fantasy_world_amd/tensor_parallel.py),
engine driven with the torch op set; every rank must reproduce the single-process result.  TP 2 and TP 4 (12 bicross heads divide:
head split + all-reduce), TP 8 (they do not: query-row split + all-gather; 5 DiT / 2 VGGT heads per rank), 2 CFG groups x TP 2
through denoise_step, row-blocked reductions, return_prediction on every rank."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cfg():
    from fantasy_world_amd import config as fwc
    return fwc.plumbing(num_layers=2, start_index=1, ffn_dim=512)      # 8 x 64-wide FFN slabs: one per rank at TP 8


def _hc():
    import dataclasses
    from fantasy_world_amd import config as fwc
    return dataclasses.replace(fwc.HeadsConfig.e2e_small(), layer_idx=[0, 0, 0, 0])


@pytest.fixture(scope="module")
def shared_weights(tmp_path_factory):
    from fantasy_world_amd import synth
    W = synth.make_weights(_cfg())
    W.update(synth.make_heads_weights(_hc()))
    path = str(tmp_path_factory.mktemp("w") / "weights.pt")
    torch.save(dict(W), path)
    return W, path


def _worker(rank, world, port, grid, outdir, wpath, cfg_parallel, opts=None):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1 if world > 4 else 2)
    from fantasy_world_amd import synth, parallel
    from fantasy_world_amd.sampler import FlowMatchScheduler, denoise_step
    from oracle.ref_ops import TorchRefOps
    dist.init_process_group("gloo", rank=rank, world_size=world)
    topo = parallel.make_topology(rank, world, cfg_parallel=cfg_parallel, mode="tp", reduce_dtype=torch.float32)
    assert topo.mode == "tp" and topo.tp.world == (world // 2 if cfg_parallel else world) and "tensor-parallel" in topo.describe()
    topo.tp.chunk_rows = 16                                   # exercise the row-blocked reductions at test sizes
    stats = parallel.enable_comm_stats()
    cfg = _cfg()
    W = torch.load(wpath, map_location="cpu", mmap=True, weights_only=True)
    ins = synth.make_inputs(cfg, *grid, seed=3)
    eng = parallel.make_engine(cfg, W.__getitem__, TorchRefOps(exact=bool(opts)), topo, heads_cfg=_hc(), **(opts or {}))
    cond = dict(clip_feature=ins["clip_feature"], y=ins["y"], plucker_fea=ins["plucker_fea"],
                plucker_context_lens=ins["plucker_context_lens"])
    if cfg_parallel:
        sched = FlowMatchScheduler()
        sched.set_timesteps(4)
        lat, _ = denoise_step(eng, sched, 1, ins["x"], ins["context"], ins["context_neg"], cond, topo=topo)
        res = (lat, None)
    else:
        res = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], return_prediction=True, **cond)
    kinds = sorted(stats.summary(1)["by_kind"])
    torch.save((res, kinds), os.path.join(outdir, f"tp_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_tensor_parallel_forward_matches_single_process(world, tmp_path, shared_weights):
    from fantasy_world_amd import synth
    from fantasy_world_amd.engine import FusionEngine
    from oracle.ref_ops import TorchRefOps
    grid = (3, 8, 8)
    cfg = _cfg()
    W, wpath = shared_weights
    ins = synth.make_inputs(cfg, *grid, seed=3)
    eng = FusionEngine(cfg, W.__getitem__, TorchRefOps(), heads_cfg=_hc())
    want, wpred = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], clip_feature=ins["clip_feature"], y=ins["y"],
                                    plucker_fea=ins["plucker_fea"], plucker_context_lens=ins["plucker_context_lens"],
                                    return_prediction=True)
    del eng
    mp.spawn(_worker, args=(world, _free_port(), grid, str(tmp_path), wpath, False), nprocs=world, join=True)
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
    for r in range(world):
        (got, pred), kinds = torch.load(os.path.join(str(tmp_path), f"tp_{r}.pt"))
        assert rel(got, want) < 1e-5, (r, rel(got, want))
        for k, v in wpred.items():
            assert pred[k].shape == v.shape and rel(pred[k], v) < 2e-5, (r, k, rel(pred[k], v))
        # the collectives of north_star's scheme: activations all-reduced, statistics all-reduced, adapter all-reduced;
        # 12 bicross heads: head split at TP 2 / 4, query-row all-gather at TP 8
        assert {"all_reduce", "all_reduce_stats", "all_reduce_adapter"} <= set(kinds), kinds
        assert ("all_gather_rows" in kinds) == (world == 8), kinds


@pytest.mark.parametrize("world", [2, 4, 8])
def test_fp8_linears_and_fp8_attention_under_tensor_parallelism(world, tmp_path, shared_weights):
    """BASELINE config 5 names TP = 8: precision="fp8" + fp8_attention under north_star's partition.  Column-parallel
    fp8 linears quantise the replicated full-K activation exactly like the unsharded engine; row-parallel ones (o, cross-attention o,
    FFN-2) hold a K-slice and take scale_a from the FULL row -- local row maxima all-reduced with MAX before quantising
    (`all_reduce_amax`) -- so every rank's e4m3 bytes are a slice of the unsharded quantised row; fp8 attention runs on the rank's heads.
    Partial sums are added in another order than the unsharded GEMM adds them, and e4m3 rounding decisions downstream amplify those
    last-bit differences: the sharded forward sits at "another run of the same arithmetic" from the unsharded fp8 engine (golden level,
    bounded at 2e-2: the fp8 path's stated tolerance), and at e4m3's own price from the exact-attention fp32 engine."""
    from fantasy_world_amd import synth
    from fantasy_world_amd.engine import FusionEngine
    from oracle.ref_ops import TorchRefOps
    grid = (3, 8, 8)
    cfg = _cfg()
    W, wpath = shared_weights
    ins = synth.make_inputs(cfg, *grid, seed=3)
    opts = dict(precision="fp8", fp8_attention=True)
    kw = dict(clip_feature=ins["clip_feature"], y=ins["y"], plucker_fea=ins["plucker_fea"], plucker_context_lens=ins["plucker_context_lens"])
    want, wpred = FusionEngine(cfg, W.__getitem__, TorchRefOps(exact=True), heads_cfg=_hc(), **opts).joint_forward(
        ins["x"], ins["timestep"], ins["context"], return_prediction=True, **kw)
    full, _ = FusionEngine(cfg, W.__getitem__, TorchRefOps(exact=True)).joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
    mp.spawn(_worker, args=(world, _free_port(), grid, str(tmp_path), wpath, False, opts), nprocs=world, join=True)
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
    price = rel(want, full)
    assert 1e-3 < price < 1e-1, price
    for r in range(world):
        (got, pred), kinds = torch.load(os.path.join(str(tmp_path), f"tp_{r}.pt"))
        assert rel(got, want) < 2e-2, (r, rel(got, want))
        assert rel(got, full) < 1.5 * price + 1e-3, (r, rel(got, full), price)
        assert torch.equal(got, torch.load(os.path.join(str(tmp_path), "tp_0.pt"))[0][0])          # every rank returns the same tensor
        for k, v in wpred.items():
            assert pred[k].shape == v.shape and rel(pred[k], v) < 5e-2, (r, k, rel(pred[k], v))
        assert {"all_reduce", "all_reduce_stats", "all_reduce_adapter", "all_reduce_amax"} <= set(kinds), kinds


def test_row_parallel_fp8_linear_uses_the_scale_of_the_full_row():
    """The arithmetic of one row-parallel fp8 linear, without processes: K cut in 4 slices, row maxima combined with MAX, every slice
    quantised with the full row's scale -> the concatenated e4m3 bytes ARE the unsharded quantised row (bit for bit) and the summed
    partial products equal the unsharded fp8 linear to fp32 round-off; quantising each slice with its OWN maximum would not (rows whose
    maximum exceeds 448 get different scales)."""
    from oracle.ref_ops import TorchRefOps
    ops = TorchRefOps(exact=True)
    g = torch.Generator().manual_seed(5)
    M, K, N, n = 64, 512, 96, 4
    x = (torch.randn(M, K, generator=g) * 300).to(torch.bfloat16).float()        # amax well above 448 on most rows: scale_a > 1
    w, b = torch.randn(N, K, generator=g) * 0.05, torch.randn(N, generator=g)
    lin = ops.pack_linear(w, b, fp8=True)
    want = ops.linear(x, lin, out_f32=True)
    q_full, s_full = ops.quantize_fp8_rows(x)
    assert float(s_full.max()) > 1.0
    amax = torch.stack([ops.row_absmax(x[:, i * K // n:(i + 1) * K // n]) for i in range(n)]).amax(dim=0)      # the MAX all-reduce
    parts, naive = [], []
    for i in range(n):
        sl = slice(i * K // n, (i + 1) * K // n)
        q, s = ops.quantize_fp8_rows(x[:, sl], amax=amax)
        assert torch.equal(q.view(torch.uint8), q_full[:, sl].view(torch.uint8)) and torch.equal(s, s_full)
        part = ops.pack_linear(w[:, sl].contiguous(), None, fp8=True)
        parts.append(ops.linear((q, s), part, out_f32=True))
        naive.append(ops.linear(x[:, sl], part, out_f32=True))
    got = sum(parts) + ops.fp8_bias(b)
    assert ((got - want).norm() / want.norm()).item() < 1e-6
    assert ((sum(naive) + ops.fp8_bias(b) - want).norm() / want.norm()).item() > 1e-3


def test_cfg_groups_times_tensor_parallel_step(tmp_path, shared_weights):
    """world 4 = 2 CFG groups x TP 2 through sampler.denoise_step: the layout `FW_PARALLEL=tp bench.py --gpus 4` runs."""
    from fantasy_world_amd import synth
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.sampler import FlowMatchScheduler, denoise_step
    from oracle.ref_ops import TorchRefOps
    grid, world = (2, 8, 8), 4
    cfg = _cfg()
    W, wpath = shared_weights
    ins = synth.make_inputs(cfg, *grid, seed=3)
    eng = FusionEngine(cfg, W.__getitem__, TorchRefOps())
    cond = dict(clip_feature=ins["clip_feature"], y=ins["y"], plucker_fea=ins["plucker_fea"],
                plucker_context_lens=ins["plucker_context_lens"])
    sched = FlowMatchScheduler()
    sched.set_timesteps(4)
    want, _ = denoise_step(eng, sched, 1, ins["x"], ins["context"], ins["context_neg"], cond)
    del eng
    mp.spawn(_worker, args=(world, _free_port(), grid, str(tmp_path), wpath, True), nprocs=world, join=True)
    for r in range(world):
        (got, _), kinds = torch.load(os.path.join(str(tmp_path), f"tp_{r}.pt"))
        err = ((got.double() - want.double()).norm() / want.double().norm()).item()
        assert err < 1e-5, (r, err)
        assert "all_gather_cfg" in kinds


def test_tensor_shard_splits():
    from fantasy_world_amd.tensor_parallel import TensorShard
    ts = [TensorShard(r, 8) for r in range(8)]
    assert [t.heads(40) for t in ts][:2] == [(0, 5), (5, 10)] and ts[7].heads(16) == (14, 16)
    assert [t.units(13824) for t in ts][0] == (0, 1728) and ts[7].units(13824) == (12096, 13824)
    assert not ts[0].divides(12) and TensorShard(0, 4).divides(12)
    a, b, c = ts[3].rows(32865)
    assert sum(c) == 32865 and b - a == c[3]
    with pytest.raises(ValueError):
        TensorShard(0, 3).heads(40)
