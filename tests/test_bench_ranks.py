"""bench.py's multi-rank path on the one GPU of the test box: two ranks over gloo sharing GPU 0
(the driver's real runs use one GPU per rank over RCCL).  The photon range is split between the
ranks; the reduced counters must equal a single rank tracing the whole range."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def run_bench(ranks, photons):
    common = ["bench.py", "--gpus", str(ranks), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--scaling", "strong",
              "--photons", str(photons)]
    if ranks == 1:
        cmd = [sys.executable] + common
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
               "--master-addr", "127.0.0.1", "--master-port", "29541"] + common + ["--backend", "gloo", "--share-device"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [line for line in r.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    return json.loads(lines[0])


def test_two_ranks_add_up_to_one():
    photons = 3_000_000
    one = run_bench(1, photons)
    two = run_bench(2, photons)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["config"]["photons_per_light"] == one["config"]["photons_per_light"] == photons
    for key in ("photons", "casts", "bounces", "deposits", "kept", "culled", "store_attempted"):
        assert two["counters"][key] == one["counters"][key], key
    assert two["value"] > 0 and two["roofline"]["kernel_ms"] > 0
