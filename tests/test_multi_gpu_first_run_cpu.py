"""First-run safety of the N > 1 path, world_size 2 over gloo on CPU:
  * the grouped q|k|v exchange is probed on a side communicator and falls back to one exchange per attention -- on EVERY rank -- when
    it does not complete in time on ANY rank (SequenceShard.negotiate_exchange_groups);
  * parallel.golden_self_check: the golden case through each rank's shard, MAX over ranks; one rank off the golden refuses the run;
  * the environment a driver-launched rank needs is set by importing the package's parallel module, the watchdog default is below
    the driver's limit and printed."""
import os
import socket
import time

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


def _init(rank, world, port):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from fantasy_world_amd import parallel
    return parallel, parallel.init_topology(backend="gloo", cfg_parallel=False)


def _probe_worker(rank, world, port, outdir, mode):
    parallel, topo = _init(rank, world, port)
    sh = topo.shard
    assert sh.probe_group is not None and sh.probe_group is not sh.group
    if mode == "late_rank" and rank == 1:
        # this rank reaches the probe 2.5 s late: rank 0's grouped exchange cannot complete within its 1 s limit
        orig = sh._probe_grouped_exchange
        sh._probe_grouped_exchange = lambda device, timeout_s: (time.sleep(2.5), orig(device, 30.0))[1]
    t0 = time.monotonic()
    ran = sh.negotiate_exchange_groups(2, "cpu", timeout_s=1.0 if mode == "late_rank" else 20.0)
    # the forward's own communicator is untouched by whatever happened on the probe communicator
    t = torch.full((4,), float(rank + 1))
    dist.all_reduce(t, group=sh.group)
    torch.save((ran, dict(sh.exchange_probe), float(t[0]), time.monotonic() - t0), os.path.join(outdir, f"probe_{rank}.pt"))
    assert sh.negotiate_exchange_groups(2, "cpu") == ran          # cached: negotiated once per shard
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["healthy", "late_rank"])
def test_grouped_exchange_probe_and_fallback(mode, tmp_path):
    world = 2
    mp.spawn(_probe_worker, args=(world, _free_port(), str(tmp_path), mode), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"probe_{r}.pt", weights_only=False) for r in range(world)]
    want = 2 if mode == "healthy" else 1
    for ran, probe, s, _ in res:
        assert ran == want and probe["requested"] == 2 and probe["ran"] == want and probe["ok"] == (mode == "healthy")
        assert s == 3.0                                             # 1 + 2: the main communicator still works
    if mode == "late_rank":
        assert res[0][3] >= 1.0                                     # rank 0 waited out its limit, then agreed with rank 1


class _NoisyOps:
    """The torch op set with one op perturbed: a rank whose kernels are wrong."""

    def __init__(self, ops, rel):
        self._ops, self._rel = ops, rel

    def __getattr__(self, name):
        return getattr(self._ops, name)

    def linear(self, x, lin, *a, **k):
        out = self._ops.linear(x, lin, *a, **k)
        return out * (1.0 + self._rel) if out is not None and out.is_floating_point() else out


def _golden_worker(rank, world, port, outdir, bad_rank):
    parallel, topo = _init(rank, world, port)
    from oracle.ref_ops import TorchRefOps
    ops = TorchRefOps()
    if rank == bad_rank:
        ops = _NoisyOps(ops, 2e-2)
    res = parallel.golden_self_check(topo, ops, case="wan21_l3_f2_12x8", tol=1e-3)
    torch.save(res, os.path.join(outdir, f"golden_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bad_rank", [-1, 1])
def test_golden_self_check_over_two_ranks(bad_rank, tmp_path):
    """2-rank sequence shard of the 3-block golden model (2 latent frames -> one per rank): every rank sees the MAX error over ranks, so a
    single bad rank makes ALL ranks refuse (bench.py then prints value = null and exits 3)."""
    world = 2
    mp.spawn(_golden_worker, args=(world, _free_port(), str(tmp_path), bad_rank), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"golden_{r}.pt", weights_only=False) for r in range(world)]
    assert res[0] == res[1]
    if bad_rank < 0:
        assert res[0]["ok"] and res[0]["rel_l2_max_over_ranks"] < 1e-4, res[0]
    else:
        assert not res[0]["ok"] and res[0]["rel_l2_max_over_ranks"] > 1e-3, res[0]


def test_first_run_environment_and_watchdog_defaults(monkeypatch):
    import importlib
    import re
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    from fantasy_world_amd import parallel
    importlib.reload(parallel)
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"          # whoever launched the ranks
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index('setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")') < main.index("import torch")
    m = re.search(r'FW_BENCH_WATCHDOG_S", "0" if .* else "(\d+)"', src)
    assert m and int(m.group(1)) <= 900 and '"watchdog_s"' in src
    monkeypatch.setenv("FW_TP_REDUCE_DTYPE", "fp16")
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises((ValueError, RuntimeError)):
        parallel.init_topology(backend="gloo")


def test_packaged_golden_is_the_repository_fixture():
    """fantasy_world_amd/golden/ ships the default case of golden_self_check inside the package: it must be the
    tests/golden fixture's meta + noise_pred, and a missing fixture must raise instead of skipping the check."""
    import pytest
    from fantasy_world_amd import parallel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    case = "wan21_cfg1_l2_f9_64x64"
    a = torch.load(os.path.join(root, "fantasy_world_amd", "golden", case + ".pt"), map_location="cpu", weights_only=False)
    b = torch.load(os.path.join(root, "tests", "golden", case + ".pt"), map_location="cpu", weights_only=False)
    assert a["meta"] == b["meta"] and torch.equal(a["noise_pred"], b["noise_pred"])
    with pytest.raises(FileNotFoundError):
        parallel.golden_self_check(None, None, case="no_such_case")
