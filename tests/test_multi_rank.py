"""The N > 1 path on CPU: two gloo ranks, each tracing its photon-index shard with the
oracle as the per-rank engine, then the same all-reduce bench.py performs over RCCL.
The sum over ranks must equal the single-rank frame: counters and store records exactly
(photon paths depend only on seed / light / photon index), float camera buffers to the
rounding of a different summation order."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from conftest import REPO

parallel = importlib.import_module("ray-tracer_amd.parallel")

SCENE = os.path.join(REPO, "tests", "scenes", "mixed.json")
PHOTONS = 1201  # odd on purpose: the shards are unequal
COUNTERS = ("photons", "casts", "bounces", "deposits", "splat_total", "kept", "culled", "store_size")


def _trace(rt, first, count, sink):
    scene = rt.Scene(SCENE)
    scene.settings.photons_per_light = PHOTONS  # the GLOBAL budget: per-photon flux is Phi / PHOTONS on every rank
    orc = O.Oracle(scene)
    kwargs = {}
    if sink == O.SINK_SPLAT:
        d = orc.derive()
        kwargs = dict(min_splat_radius=d["min_splat_radius"], density_cell_size=d["cell_size"])
    else:
        probes = orc.collect_probe_positions()
        kwargs = dict(probes=probes, probe_cell_size=6.0, keep_radius=6.0, store_capacity=1 << 16)
    return orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, sink_mode=sink, first_photon=first,
                     photon_count=count, **kwargs)


def _rank_main(rank, world, port, sink, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rt = importlib.import_module("ray-tracer_amd")
    first, count = parallel.shard_range(PHOTONS, rank, world)
    r = _trace(rt, first, count, sink)
    buffers = [torch.from_numpy(b.copy()) for b in r["splat"]]
    totals = torch.tensor([r[k] for k in COUNTERS], dtype=torch.int64)
    parallel.all_reduce_sum(buffers + [totals], dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, r["store"])
    if rank == 0:
        np.savez(os.path.join(out_dir, "reduced.npz"), totals=totals.numpy(),
                 store=np.concatenate(gathered, axis=0), **{f"cam{i}": b.numpy() for i, b in enumerate(buffers)})
    dist.destroy_process_group()


def _sorted(store):
    return store[np.lexsort(store.view(np.uint32).T[::-1])]


@pytest.mark.parametrize("sink", [O.SINK_SPLAT, O.SINK_PROBE])
def test_two_ranks_equal_one(rt, tmp_path, sink):
    world = 2
    port = 29500 + os.getpid() % 2000 + sink
    mp.spawn(_rank_main, args=(world, port, sink, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "reduced.npz")
    whole = _trace(rt, 0, PHOTONS, sink)
    assert [int(v) for v in got["totals"]] == [whole[k] for k in COUNTERS]
    if sink == O.SINK_SPLAT:
        assert len(whole["splat"]) >= 1 and whole["splat_total"] > 0
        for i, want in enumerate(whole["splat"]):
            np.testing.assert_allclose(got[f"cam{i}"], want, rtol=1e-5, atol=1e-7 * float(np.abs(want).max()))
    else:
        assert whole["store_size"] > 0
        assert np.array_equal(_sorted(got["store"]).view(np.uint32), _sorted(whole["store"]).view(np.uint32))


def test_shard_ranges_tile_the_budget():
    for photons in (0, 1, 7, 1201, 200_000_000):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(photons, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == photons
            for (a, n), (b, _) in zip(spans, spans[1:]):
                assert a + n == b
    with pytest.raises(ValueError):
        parallel.shard_range(10, 2, 2)


def test_store_shares_add_up_to_the_one_store():
    """A sharded frame has the reference's ONE store (src/Renderer.cpp:382-390): rank 0 keeps the emitter
    deposits, and the ranks' equal shares of the rest add up to it (rounded up per rank; a share is at
    least one record, since a limit of 0 means the whole store)."""
    for capacity, emitters in ((7_548_416, 0), (40_000_000, 12_345_678), (1000, 999), (10, 25), (1, 0)):
        for world in (1, 2, 3, 8):
            shares = [parallel.store_share(capacity, emitters, r, world) for r in range(world)]
            kept_emitters = min(emitters, capacity)
            part = -(-(capacity - kept_emitters) // world)
            assert shares == [max(1, (kept_emitters if r == 0 else 0) + part) for r in range(world)]
            assert capacity <= sum(shares) <= capacity + 2 * world
            if world == 1:
                assert shares == [capacity]
    with pytest.raises(ValueError):
        parallel.store_share(10, 0, 2, 2)
