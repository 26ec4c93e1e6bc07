"""A whole frame sharded over two ranks (gloo, both on the test box's one GPU) equals the frame of
one rank: same counters, camera buffers equal up to the order of float sums.  Probe-gather mode
(per-rank store and gather, emitter deposits on rank 0 only) and splat mode."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu
WORKER = os.path.join(REPO, "tests", "frame_ranks_worker.py")


def render(ranks, scene, photons, probe_gather, out, store_capacity=None):
    args = [WORKER, os.path.join(REPO, scene), str(photons), "1" if probe_gather else "0", out]
    if store_capacity is not None:
        args.append(str(store_capacity))
    if ranks == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
               "--master-addr", "127.0.0.1", "--master-port", "29547"] + args
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    return np.load(out + ".npz"), json.load(open(out + ".json"))


@pytest.mark.parametrize("probe_gather", [True, False])
@pytest.mark.parametrize("scene,photons", [("tests/scenes/mixed.json", 30_001), ("scenes/MultiCameraTest.json", 200_000)])
def test_two_rank_frame_equals_one_rank_frame(tmp_path, scene, photons, probe_gather):
    one, c1 = render(1, scene, photons, probe_gather, str(tmp_path / "one"))
    two, c2 = render(2, scene, photons, probe_gather, str(tmp_path / "two"))
    for k in ("photons", "casts", "bounces", "deposits", "kept", "culled", "splat_total"):
        assert c1[k] == c2[k], k
    assert c1["bounces"] > 0
    lit = 0
    for name in one.files:
        a, b = one[name], two[name]
        peak = float(np.abs(a).max())
        lit += peak > 0
        # sums of the same terms in another order
        np.testing.assert_allclose(b, a, rtol=3e-4, atol=2e-6 * max(peak, 1e-30))
    assert lit > 0


def test_more_ranks_than_photons(tmp_path):
    """One photon per light over two ranks: rank 0's share is empty and must trace nothing (a photon
    count of 0 in rt_trace_params means "all the rest"), so the frame holds each photon once."""
    one, c1 = render(1, "tests/scenes/mixed.json", 1, True, str(tmp_path / "one"))
    two, c2 = render(2, "tests/scenes/mixed.json", 1, True, str(tmp_path / "two"))
    assert c1["photons"] == c2["photons"] == 5  # one per light
    for k in ("casts", "bounces", "deposits", "kept"):
        assert c1[k] == c2[k], k


def test_overflowing_store_keeps_as_many_records_on_two_ranks(tmp_path):
    """The frame has ONE bounce store (reference src/Renderer.cpp:382-390, src/BounceStore.cpp:21-33: appends
    beyond its capacity are dropped): sharded over two ranks every rank keeps its share of it, so the
    frame holds as many records, and as much light, as the frame of one rank.  (With a whole store per
    rank the two-rank frame would hold twice the records and come out twice as bright.)"""
    scene, photons, capacity = "scenes/MultiCameraTest.json", 200_000, 20_000
    one, c1 = render(1, scene, photons, True, str(tmp_path / "one"), capacity)
    two, c2 = render(2, scene, photons, True, str(tmp_path / "two"), capacity)
    assert c1["store_attempted"] == c2["store_attempted"] > 2 * capacity  # both overflow, by a wide margin
    assert c1["store_size"] == capacity
    assert capacity <= c2["store_size"] <= capacity + 2
    lit = 0
    for name in one.files:
        a, b = one[name].astype(np.float64), two[name].astype(np.float64)
        if a.sum() <= 0:
            continue
        lit += 1
        # which records are kept differs (the first to arrive on each GPU), how many does not
        assert abs(b.sum() / a.sum() - 1.0) < 0.15, (name, a.sum(), b.sum())
    assert lit > 0
