"""bench.py's multi-rank path on the one GPU of the test box: two ranks over gloo sharing GPU 0
(the driver's real runs use one GPU per rank over RCCL).  The photon range is split between the
ranks; the reduced counters must equal a single rank tracing the whole range."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO



def run_bench(ranks, photons, launcher=True, extra=()):
    common = ["bench.py", "--gpus", str(ranks), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--scaling", "strong",
              "--photons", str(photons)] + list(extra)
    if ranks == 1:
        cmd = [sys.executable] + common
    elif launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
               "--master-addr", "127.0.0.1", "--master-port", "29541"] + common + ["--backend", "gloo", "--share-device"]
    else:  # bench.py starts its ranks itself
        cmd = [sys.executable] + common + ["--backend", "gloo", "--share-device"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [line for line in r.stdout.splitlines() if line.strip()]
    assert len(lines) == 1, "standard output is exactly one JSON line, from rank 0: " + r.stdout[:400]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_ranks_add_up_to_one():
    """The whole-frame step (emitter deposits on rank 0, pass, per-rank gather, all-reduce): counters of
    two ranks equal one rank's, and so does the gathered image up to float summation order."""
    photons = 3_000_000
    one = run_bench(1, photons)
    two = run_bench(2, photons)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["config"]["photons_per_light"] == one["config"]["photons_per_light"] == photons
    assert "gather" in two["config"]["step"] and "all-reduce" in two["config"]["step"]
    for key in ("photons", "casts", "bounces", "deposits", "kept", "culled", "store_attempted"):
        assert two["counters"][key] == one["counters"][key], key
    assert two["value"] > 0 and two["roofline"]["kernel_ms"] > 0
    assert two["stages_ms"]["gather"] > 0


@pytest.mark.gpu
def test_gpus_without_a_launcher_starts_the_ranks():
    """`python bench.py --gpus 2` with no launcher around it must not report one GPU's number as two."""
    line = run_bench(2, 400_000, launcher=False)
    assert line["n_gpus"] == 2 and line["counters"]["photons"] == 400_000


@pytest.mark.gpu
def test_one_rank_over_rccl():
    """The RCCL branch itself: one rank, process group on the nccl backend, the step's all-reduces
    running over it."""
    plain = run_bench(1, 400_000)
    rccl = run_bench(1, 400_000, extra=["--collective", "always", "--backend", "nccl"])
    assert "all-reduce" in rccl["config"]["step"] and "all-reduce" not in plain["config"]["step"]
    for key in ("photons", "casts", "bounces", "deposits", "kept", "store_size"):
        assert rccl["counters"][key] == plain["counters"][key], key


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", [True, False])
def test_a_rank_that_dies_ends_the_job_with_a_failure(launcher):
    """One rank of two exits before its first step: the job (under torch.distributed.run, or started by
    bench.py itself) ends with a non-zero status within the time-out instead of waiting in a collective for the
    rank that is gone, and prints no result line."""
    import time
    common = ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--photons", "200000",
              "--backend", "gloo", "--share-device"]
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29547"] + common
    else:
        cmd = [sys.executable] + common
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", RT_BENCH_FAIL_RANK="1")
    started = time.monotonic()
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode != 0
    assert time.monotonic() - started < 200
    assert not [line for line in r.stdout.splitlines() if line.startswith("{")]
    assert "injected failure" in r.stdout + r.stderr


def test_gpus_must_match_the_world_size():
    """A rank count that contradicts the launcher's is an error, on any machine (no GPU is touched)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"], cwd=REPO, capture_output=True, text=True, timeout=120,
                       env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)
