"""bench.py must be runnable the way the driver runs it: `python bench.py --gpus N ...` with no torchrun environment
self-launches N ranks under torch.distributed.run on 127.0.0.1 and prints ONE JSON line from rank 0.  Proved here without a
GPU through --dry-run (rendezvous over gloo, the CFG groups, every collective of the sequence shard on small CPU tensors,
barrier + max-over-ranks timing, the `comm` block); the engine-level sharded arithmetic is covered by
tests/test_sequence_shard_cpu.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, extra=(), env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FW_PARALLEL")}
    env["OMP_NUM_THREADS"] = "1"
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
                        "--dry-run", *extra], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [1, 2, 4, 8])
def test_bench_self_launches_n_ranks(n):
    out = _run(n)
    assert out["n_gpus"] == n and out["dry_run"] is True and out["value"] is None
    assert out["steps"] == 2 and out["warmup"] == 1
    par = out["config"]["parallelism"]
    if n == 1:
        assert par == "single GPU"
    else:
        assert "CFG-parallel x2" in par
        assert ("sequence-sharded x%d" % (n // 2) in par) == (n > 2)
        kinds = out["comm"]["by_kind"]
        assert "all_gather_cfg" in kinds
        if n > 2:
            assert {"all_to_all_qkv", "all_to_all_out", "all_gather_rows"} <= set(kinds)
            assert out["comm"]["bytes_sent_per_gpu_per_step"] > 0


def test_bench_refuses_mismatched_world(monkeypatch):
    """Launched under torchrun with a different world size than --gpus: a clear error, not a hang."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True,
                       text=True, timeout=120, env=env, cwd=ROOT)
    # WORLD_SIZE is set, so no self-launch; the dry run reports the world it really has
    assert p.returncode == 0 and '"n_gpus": 1' in p.stdout


def test_bench_tensor_parallel_mode_by_environment():
    """FW_PARALLEL=tp selects north_star's partition (2 CFG groups x TP n/2) with the same command line."""
    out = _run(8, env_extra={"FW_PARALLEL": "tp"})
    par = out["config"]["parallelism"]
    assert "CFG-parallel x2" in par and "tensor-parallel x4" in par and "sequence-sharded" not in par
    assert {"all_reduce", "all_gather_rows", "all_gather_cfg"} <= set(out["comm"]["by_kind"])


@pytest.mark.parametrize("mode", ["sp", "tp"])
def test_bench_alt_block_runs_the_other_partition_in_the_same_process_group(mode):
    """N > 1: after the headline loop the OTHER partition (tensor-parallel when the headline is the CFG x sequence-shard default, and vice
    versa) runs over the same ranks and lands in an `alt` block beside the headline; dry run = its process groups
    and collectives."""
    out = _run(8, env_extra={"FW_PARALLEL": mode})
    par, alt = out["config"]["parallelism"], out["alt"]
    assert ("sequence-sharded x4" in par) == (mode == "sp") and ("tensor-parallel x4" in par) == (mode == "tp")
    assert "CFG-parallel x2" in alt["parallelism"]
    assert ("tensor-parallel x4" in alt["parallelism"]) == (mode == "sp") and ("sequence-sharded x4" in alt["parallelism"]) == (mode == "tp")
    kinds = set(alt["comm"]["by_kind"])
    assert "all_gather_cfg" in kinds and (("all_reduce" in kinds) if mode == "sp" else ("all_gather_rows" in kinds))


def test_bench_alt_block_absent_where_both_partitions_are_the_unsharded_forward():
    assert "alt" not in _run(2)            # two CFG groups of one GPU each
    assert "alt" not in _run(4, env_extra={"FW_BENCH_ALT": "0"})


def test_bench_alt_block_cannot_cost_the_headline_line():
    """A second partition that never finishes: the guard prints the headline line with alt = the error and every rank exits 0."""
    out = _run(4, env_extra={"FW_BENCH_ALT_FORCE_HANG": "1", "FW_BENCH_ALT_BUDGET_S": "3"})
    assert out["n_gpus"] == 4 and "did not finish" in out["alt"]["error"] and "comm" in out


def test_bench_matrix_pipe_blocks_are_stored_measurements_with_provenance():
    """`roofline.matrix_pipe` / `kernels.*.matrix_pipe` (round 5, replaces the `roofline_cap` model the round-4 counters contradicted) are
    STORED counter measurements: every head size the headline runs has one, it names the profile it comes from, and the fraction it
    predicts is busy x clock / 2.4 GHz."""
    sys.path.insert(0, ROOT)
    import bench
    for hd in (128, 96, 64):
        m = bench.attention_measured(hd)
        assert m is not None and 0.3 < m["mfma_busy"] < 1.0 and 1.0 < m["sustained_ghz"] < 2.4
        assert "stored PMC pass" in m["source"] and "profiles/r0" in m["source"]
        assert abs(m["predicted_frac_of_2p5_pf"] - m["mfma_busy"] * m.get("algorithmic_share_of_mfma_cycles", 1.0) * m["sustained_ghz"] / 2.4) < 1e-3
    assert bench.attention_measured(80) is None          # no measurement, no number


def test_bench_parallel_flag_selects_the_partition_without_an_environment_variable():
    """`--parallel tp` (a driver that cannot set environment variables can still choose north_star's partition),
    and it wins over $FW_PARALLEL; the line carries the PREDICTED step time of both partitions next to whatever gets measured."""
    out = _run(8, extra=("--parallel", "tp"))
    par = out["config"]["parallelism"]
    assert "tensor-parallel x4" in par and "sequence-sharded" not in par and "sequence-sharded x4" in out["alt"]["parallelism"]
    out = _run(4, extra=("--parallel", "sp"), env_extra={"FW_PARALLEL": "tp"})
    assert "sequence-sharded x2" in out["config"]["parallelism"] and "tensor-parallel x2" in out["alt"]["parallelism"]
    for blk, mode in ((out["predicted"], "sp"), (out["alt"]["predicted"], "tp")):
        assert "PREDICTION" in blk["source"] and blk["step_ms_best"] <= blk["step_ms_worst"]
    assert out["predicted"]["step_ms_worst"] < out["alt"]["predicted"]["step_ms_best"]        # what the default was chosen on


def test_bench_alt_guard_scales_with_what_the_run_measured(monkeypatch):
    """The budget of the second partition's block: 420 s at the headline's predicted step times, more when the run's own engine build
    and step time say so (config 4 / 5: two experts at 720p), an explicit $FW_BENCH_ALT_BUDGET_S wins."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.delenv("FW_BENCH_ALT_BUDGET_S", raising=False)
    assert bench.alt_budget_s(40.0, 500.0, 5, 1) == 420.0
    big = bench.alt_budget_s(300.0, 3000.0, 5, 2)
    assert big == 120.0 + 1200.0 + 12 * 7 * 3.0 and big > 420.0
    monkeypatch.setenv("FW_BENCH_ALT_BUDGET_S", "77")
    assert bench.alt_budget_s(300.0, 3000.0, 5, 2) == 77.0
    assert bench.predicted_block(8, "sp", True)["step_ms_worst"] == 521 and bench.predicted_block(8, "sp", False) is None
    assert bench.predicted_block(3, "sp", True) is None
