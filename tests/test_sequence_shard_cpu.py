"""N > 1 path on CPU: world_size 2 and 3 over gloo, engine driven with the torch op set (test infrastructure); every rank
must reproduce the single-process result.  Covers the row/frame split (uneven frames), K/V all-gathers, the bridge
re-shard and the head gather."""
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
    return fwc.plumbing(num_layers=2, start_index=1, ffn_dim=256)


def _worker(rank, world, port, grid, outdir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from fantasy_world_amd import synth
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.parallel import SequenceShard
    from oracle.ref_ops import TorchRefOps
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = _cfg()
    W = synth.make_weights(cfg)
    ins = synth.make_inputs(cfg, *grid, seed=3)
    eng = FusionEngine(cfg, W.__getitem__, TorchRefOps(), shard=SequenceShard(rank, world))
    out, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], clip_feature=ins["clip_feature"], y=ins["y"],
                               plucker_fea=ins["plucker_fea"], plucker_context_lens=ins["plucker_context_lens"])
    torch.save(out, os.path.join(outdir, f"out_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,grid", [(2, (3, 8, 8)), (3, (4, 4, 12))])
def test_sequence_shard_matches_single_process(world, grid, tmp_path):
    from fantasy_world_amd import synth
    from fantasy_world_amd.engine import FusionEngine
    from oracle.ref_ops import TorchRefOps
    cfg = _cfg()
    W = synth.make_weights(cfg)
    ins = synth.make_inputs(cfg, *grid, seed=3)
    eng = FusionEngine(cfg, W.__getitem__, TorchRefOps())
    want, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], clip_feature=ins["clip_feature"], y=ins["y"],
                                plucker_fea=ins["plucker_fea"], plucker_context_lens=ins["plucker_context_lens"])
    del eng, W
    mp.spawn(_worker, args=(world, _free_port(), grid, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), f"out_{r}.pt"))
        err = ((got.double() - want.double()).norm() / want.double().norm()).item()
        assert err < 1e-5, (r, err)


def test_split_counts():
    from fantasy_world_amd.parallel import split_counts
    assert split_counts(21, 8) == [3, 3, 3, 3, 3, 2, 2, 2]
    assert split_counts(32760, 8) == [4095] * 8
    assert sum(split_counts(75600, 8)) == 75600
