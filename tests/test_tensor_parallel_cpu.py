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


def _worker(rank, world, port, grid, outdir, wpath, cfg_parallel):
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
    eng = parallel.make_engine(cfg, W.__getitem__, TorchRefOps(), topo, heads_cfg=_hc())
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
