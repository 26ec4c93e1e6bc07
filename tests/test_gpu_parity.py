"""HIP path vs the reference's golden vectors and vs the CPU oracle (MI355X only).

Everything here goes through the C-ABI of librt_amd.so.  Bar: bit-exact for hits,
ids, bounce counts, photon power and store records; camera buffers (float atomics,
order-dependent) to 1e-5 relative.
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import REPO, assert_bits_equal, assert_one_more_frame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tracer_mod():
    from importlib import import_module
    return import_module("ray-tracer_amd.tracer")


UNIT = [
    ("tri", "rt_intersect_triangles", 8, np.float64),
    ("bounds", "rt_intersect_bounds", 0, np.int32),
    ("sphere", "rt_intersect_spheres", 8, np.float64),
    ("quad", "rt_intersect_quads", 20, np.float64),
    ("plane", "rt_intersect_planes", 8, np.float64),
    ("eval", "rt_bsdf_evaluate", 4, np.float64),
]


@pytest.mark.parametrize("name,fn,cols,dtype", UNIT)
def test_device_intersectors_match_reference_bits(tracer_mod, golden, name, fn, cols, dtype):
    g = golden("unit")
    got = tracer_mod.rows(fn, g[f"{name}_in"], cols, dtype)
    assert_bits_equal(got, g[f"{name}_out"].astype(dtype), name)


def test_device_bsdf_sample(tracer_mod, golden):
    """Mirror and dielectric use no transcendental: bit-exact vs the reference.  Lambert and
    microfacet go through the shared sin/cos: bit-exact vs the oracle in portable mode, and
    within a few ulp of the reference."""
    g = golden("unit")
    got = tracer_mod.rows("rt_bsdf_sample", g["bsdf_in"], 10)
    ref = g["bsdf_out"]
    delta = np.isin(g["bsdf_in"][:, 0], (1, 2))
    assert_bits_equal(got[delta], ref[delta], "delta materials")
    portable = O.batch("orc_bsdf_sample", g["bsdf_in"], 10, extra=(O.C.c_int32(O.MATH_PORTABLE),))
    assert_bits_equal(got, portable, "all materials vs oracle")
    assert np.array_equal(got[:, 0], ref[:, 0])
    assert np.abs(got[:, 1:8] - ref[:, 1:8]).max() < 1e-14


def _open(rt, scene_opener, entry):
    scene = scene_opener(entry)
    return scene, rt.Tracer(scene), O.Oracle(scene)


def test_device_emission_matches_oracle_and_reference(rt, golden, manifest, scene_opener):
    entry = manifest["emit"]
    g = golden("emit")
    scene, tracer, orc = _open(rt, scene_opener, entry)
    n = entry["n"]
    draws_per_photon = {0: 5, 1: 5, 2: 3, 3: 4}
    for light in range(scene.desc.light_count):
        # rebuild the 5-per-photon draw tape from the reference's sequential stream
        desc = scene.desc.lights[light]
        k = draws_per_photon[desc.type]
        if desc.type in (0, 1) and desc.inner_radius <= 0:
            k = 2
        if desc.type == 2 and desc.radius <= 0:
            k = 0
        tape = np.zeros((n, 5))
        if k:
            tape[:, :k] = g[f"emit_L{light}_draws"][:n * k].reshape(n, k)
        rays, colors = tracer.emit(light, tape, n)
        o_rays, o_colors = orc.emit(light, tape, n, O.MATH_PORTABLE)
        assert_bits_equal(rays, o_rays, f"light {light} rays vs oracle")
        assert_bits_equal(colors, o_colors, f"light {light} colors vs oracle")
        assert_bits_equal(colors, g[f"emit_L{light}_color"], f"light {light} colors vs reference")
        assert np.abs(rays - g[f"emit_L{light}_rays"]).max() < 1e-12  # sin/cos differ by <= 1 ulp


def test_closest_hit_on_reference_rays(rt, golden, manifest, scene_opener):
    """Cast the reference walk's own rays: hit position/normal/distance bits, volume,
    triangle id and material equal the reference's (north star: bit-exact hit ids)."""
    for key in ("walk_furnace", "walk_mixed", "walk_singlephoton", "walk_mirrordirect", "walk_cbgmd",
                "walk_stochastic128"):
        g = golden(key)
        scene, tracer, orc = _open(rt, scene_opener, manifest[key])
        f64, i32 = g["walk_bounce_f64"], g["walk_bounce_i32"]
        origins = np.zeros((f64.shape[0], 3))
        # a bounce's ray starts at the previous bounce's position (or the emitted origin)
        emitted = g["walk_emitted"]
        first = i32[:, 1] == 0
        photon_row = i32[:, 0]
        origins[first] = emitted[photon_row[first], :3]
        origins[~first] = f64[np.where(~first)[0] - 1, :3]
        rays = np.concatenate([origins, f64[:, 7:10]], axis=1)
        got_f, got_i = tracer.cast_rays(rays)
        assert (got_i[:, 0] == 1).all()
        assert_bits_equal(got_f, f64[:, :7], f"{key} hits")
        assert np.array_equal(got_i[:, 1], i32[:, 2]) and np.array_equal(got_i[:, 3], i32[:, 4])
        unambiguous = i32[:, 5] <= 1
        assert np.array_equal(got_i[unambiguous, 2], i32[unambiguous, 3]), "triangle ids"
        o_f, o_i = orc.cast_rays(rays)
        assert_bits_equal(got_f, o_f, f"{key} vs oracle")
        assert np.array_equal(got_i, o_i)


def test_closest_hit_with_keyframed_volumes(rt, scene_opener):
    """rt_cast_rays poses keyframed volumes at the frame time: same hits as the oracle at
    several frame times, and the hits do move between them."""
    seen = []
    for frame_time in (0.0, 0.3, 0.85):
        entry = {"scene": "tests/scenes/animated.json", "photons_per_light": 2000, "frame_time": frame_time}
        scene, tracer, orc = _open(rt, scene_opener, entry)
        pre = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=8, record_photons=2000)
        rays = pre["emitted"][:, :6].copy()
        got_f, got_i = tracer.cast_rays(rays)
        o_f, o_i = orc.cast_rays(rays)
        assert (got_i[:, 0] == 1).sum() > 1000
        assert_bits_equal(got_f, o_f, f"t={frame_time} hits vs oracle")
        assert np.array_equal(got_i, o_i)
        seen.append(got_f)
    assert not np.array_equal(seen[0], seen[1]) and not np.array_equal(seen[1], seen[2])


@pytest.mark.parametrize("name", ["scenes/CornellBoxGlassMirrorDiffuse.json", "scenes/MirrorKnot4k.json",
                                  "scenes/CornellBox.json"])
def test_closest_hit_where_flat_meshes_meet(rt, scene_opener, name, monkeypatch):
    """A room's walls are flat meshes of one pose: the device casts them together through one tree over all
    their triangles, where the reference casts volume after volume and keeps the first of equally distant
    hits (reference src/Worker.cpp:539-567).  The cases that could tell the two apart are rays aimed exactly
    at the lines and corners where walls meet (two volumes hit at one point), rays that start ON a wall (the
    wall's own hit lies below the self-hit floor and the wall must then contribute nothing) and both floors
    in use (1e-4 of the photon pass, epsilon of the probe pass).  Hits, distances and volume / triangle /
    material ids equal the oracle's bit for bit, and equal those of a tracer that casts every wall on its own
    (RT_AMD_GROUP_FLAT_MESHES=0)."""
    entry = {"scene": name, "photons_per_light": 1000}
    scene, tracer, orc = _open(rt, scene_opener, entry)
    rng = np.random.default_rng(5)
    lo, hi = np.array([-150.0, 0.0, -150.0]), np.array([150.0, 300.0, 150.0])  # the room of the Cornell scenes
    n = 4000
    inside = lo + (hi - lo) * rng.uniform(0.05, 0.95, size=(n, 3))
    # targets on the room's edges and corners: every coordinate at a wall or a round number in between
    grid = np.stack([rng.choice([-150.0, 150.0, -75.0, 0.0, 37.5], size=n), rng.choice([0.0, 300.0, 150.0, 62.5], size=n),
                     rng.choice([150.0, -150.0, 0.0, 25.0], size=n)], axis=1)
    rays = [np.concatenate([inside, grid - inside], axis=1)]
    # rays that start on a wall, leaving it, grazing along it, and heading back into it
    on_wall = inside.copy()
    axis = rng.integers(0, 3, size=n)
    side = rng.integers(0, 2, size=n)
    on_wall[np.arange(n), axis] = np.where(side == 1, hi[axis], lo[axis])
    directions = rng.normal(size=(n, 3))
    directions[: n // 4, :] = 0.0
    directions[np.arange(n // 4), (axis[: n // 4] + 1) % 3] = 1.0  # along the wall
    rays.append(np.concatenate([on_wall, directions], axis=1))
    rays = np.concatenate(rays, axis=0)
    for floor in (1e-4, 2.220446049250313e-16):
        got_f, got_i = tracer.cast_rays(rays, floor)
        o_f, o_i = orc.cast_rays(rays, floor)
        assert (got_i[:, 0] == 1).sum() > n
        assert_bits_equal(got_f, o_f, f"hits at floor {floor}")
        assert np.array_equal(got_i, o_i), f"ids at floor {floor}"
    monkeypatch.setenv("RT_AMD_GROUP_FLAT_MESHES", "0")
    apart = rt.Tracer(scene)
    a_f, a_i = apart.cast_rays(rays)
    t_f, t_i = tracer.cast_rays(rays)
    assert_bits_equal(t_f, a_f, "walls cast together and apart")
    assert np.array_equal(t_i, a_i)


SCENES = {
    "furnace": {"scene": "tests/scenes/furnace.json", "photons_per_light": 4000},
    "mixed": {"scene": "tests/scenes/mixed.json", "photons_per_light": 1500},
    "cbgmd": {"scene": "scenes/CornellBoxGlassMirrorDiffuse.json", "photons_per_light": 6000},
    "bombardment": {"scene": "scenes/MirrorBombardment-200M.json", "photons_per_light": 20000},
    "knot": {"scene": "scenes/MirrorKnot4k.json", "photons_per_light": 5000},
    "multicam": {"scene": "scenes/MultiCameraTest.json", "photons_per_light": 5000},
    "singlephoton": {"scene": "scenes/SinglePhotonVerify.json", "photons_per_light": 5000},
    "mirrordirect": {"scene": "scenes/MirrorDirectTest.json", "photons_per_light": 5000},
    "stochastic128": {"scene": "scenes/CBGMD_stochastic128.json", "photons_per_light": 6000},
    # keyframed plate, ball, knot and camera; photons are time-stamped across the shutter
    "animated": {"scene": "tests/scenes/animated.json", "photons_per_light": 3000, "frame_time": 0.3},
}


@pytest.mark.parametrize("name", sorted(SCENES))
def test_photon_paths_bit_exact_vs_oracle(rt, scene_opener, name):
    """Every photon's emitted ray, bounce count, per-bounce hit (bits), volume, triangle id,
    material and carried power equal the oracle's with the same Philox stream."""
    entry = SCENES[name]
    scene, tracer, orc = _open(rt, scene_opener, entry)
    total = entry["photons_per_light"] * scene.desc.light_count
    tracer.reset()
    tracer.trace(sink_mode=0, record_photons=total)
    got = tracer.records()
    counters = tracer.counters()
    ref = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=8, record_photons=total)
    assert_bits_equal(got["emitted"], ref["emitted"], "emitted")
    assert np.array_equal(got["casts_valid"], ref["casts_valid"]), "casts / bounce counts"
    assert_bits_equal(got["bounce_f64"], ref["bounce_f64"], "hits")
    assert np.array_equal(got["bounce_i32"], ref["bounce_i32"]), "ids"
    assert_bits_equal(got["bounce_power"], ref["bounce_power"], "power")
    for k in ("photons", "casts", "bounces", "deposits"):
        assert counters[k] == ref[k], k
    assert np.allclose(counters["deposited_power"], ref["deposited_power"], rtol=1e-9)


@pytest.mark.parametrize("name,photons", [("bombardment", 4_000_000), ("knot", 5_000_000), ("cbgmd", 1_500_000)])
def test_counters_equal_oracle_at_scale(rt, scene_opener, name, photons):
    """The device walks its own trees (surface-area splits, padded single-precision boxes, several
    triangles per leaf) and must still report the hits of the reference's exhaustive walk of its
    median-split trees, which is what the oracle runs: over this many casts every hit-or-miss
    decision and every bounce count has to agree, so the pass counters are compared exactly."""
    entry = dict(SCENES[name], photons_per_light=photons)
    scene, tracer, orc = _open(rt, scene_opener, entry)
    tracer.reset()
    tracer.trace(sink_mode=0, record_photons=20_000)
    c = tracer.counters()
    got = tracer.records()
    ref = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=8, record_photons=20_000)
    for k in ("photons", "casts", "bounces", "deposits"):
        assert c[k] == ref[k], k
    assert c["bounces"] > photons // 2
    assert_bits_equal(got["bounce_f64"], ref["bounce_f64"], "hits")
    assert np.array_equal(got["bounce_i32"], ref["bounce_i32"]), "ids"
    assert np.allclose(c["deposited_power"], ref["deposited_power"], rtol=1e-9)


@pytest.mark.parametrize("level_sort", ["default", "0", "0x134"])
@pytest.mark.parametrize("capacity", [1 << 30, 700])
@pytest.mark.parametrize("name", ["cbgmd", "knot", "mixed", "animated", "furnace"])
def test_bounce_level_queues_change_nothing(rt, scene_opener, name, capacity, level_sort, monkeypatch):
    """Large passes park the photons that go on after a bounce and trace the next bounce in a
    launch of its own, walking the parked photons in the order of a key of their rays (origin cell,
    direction sector); a full queue leaves a photon in its lane.  Forced onto small passes here
    (queue minimum 1, every level sorted), with room for everything and with room for 700 photons per
    level, in the default key layout, without the sort and with another layout: paths, counters and
    the store must equal the oracle's as in a single launch."""
    monkeypatch.setenv("RT_AMD_QUEUE_MIN", "1")
    monkeypatch.setenv("RT_AMD_LEVEL_SORT_MIN", "1")
    if level_sort != "default":
        monkeypatch.setenv("RT_AMD_LEVEL_SORT", level_sort)
    monkeypatch.setenv("RT_AMD_QUEUE_CAPACITY", str(capacity))
    entry = SCENES[name]
    scene, tracer, orc = _open(rt, scene_opener, entry)
    total = entry["photons_per_light"] * scene.desc.light_count
    pre = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=8, record_photons=600)
    probes = pre["bounce_f64"][:, 0, :3][pre["casts_valid"][:, 1] > 0][::3]
    keep = 12.0
    ref = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=8, sink_mode=O.SINK_PROBE, probes=probes,
                    probe_cell_size=keep, keep_radius=keep, store_capacity=1 << 18, record_photons=total)
    tracer.set_probes(probes, keep, keep, 1 << 18)
    tracer.reset()
    tracer.trace(sink_mode=1, record_photons=total)
    got = tracer.records()
    c = tracer.counters()
    assert ref["bounces"] > ref["photons"], "the scene must have photons that go on"
    for k in ("photons", "casts", "bounces", "deposits", "kept", "culled", "store_size"):
        assert c[k] == ref[k], k
    assert np.array_equal(got["casts_valid"], ref["casts_valid"]), "casts / bounce counts"
    assert_bits_equal(got["bounce_f64"], ref["bounce_f64"], "hits")
    assert np.array_equal(got["bounce_i32"], ref["bounce_i32"]), "ids"
    assert_bits_equal(got["bounce_power"], ref["bounce_power"], "power")
    assert_bits_equal(_sorted_records(tracer.store()), _sorted_records(ref["store"]), "store records")


def _sorted_records(store):
    order = np.lexsort(store.view(np.uint32).T[::-1])
    return store[order]


@pytest.mark.parametrize("name", ["mixed", "cbgmd", "bombardment", "animated", "stochastic128"])
def test_probe_sink_store_equals_oracle(rt, scene_opener, name):
    """Sink A: keep-test + BounceStore.  The store is a multiset (append order is a race in
    the reference too with > 1 worker): same records bit for bit after sorting; same counters."""
    entry = SCENES[name]
    scene, tracer, orc = _open(rt, scene_opener, entry)
    # probes: the first-bounce positions of a coarse subset of photons (a stand-in for the
    # camera probe pass: any point set exercises the keep-test identically on both sides)
    pre = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=8, record_photons=600)
    probes = pre["bounce_f64"][:, 0, :3][pre["casts_valid"][:, 1] > 0][::3]
    keep = 12.0
    capacity = 1 << 16
    ref = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=8, sink_mode=O.SINK_PROBE,
                    probes=probes, probe_cell_size=keep, keep_radius=keep, store_capacity=capacity)
    tracer.set_probes(probes, keep, keep, capacity)
    tracer.reset()
    tracer.trace(sink_mode=1)
    c = tracer.counters()
    assert ref["kept"] > 0 and ref["culled"] > 0
    for k in ("kept", "culled", "store_attempted", "store_size", "store_dropped", "deposits"):
        assert c[k] == ref[k], k
    assert_bits_equal(_sorted_records(tracer.store()), _sorted_records(ref["store"]), "store records")
    # overflow: capacity below the kept count drops the excess and reports it.  WHICH records a full
    # store holds is decided by the order the reference's workers meet the photons in — index order,
    # light by light (reference src/Worker.cpp:429-505, src/BounceStore.cpp:11-33) —, which is the order
    # of the oracle's single thread: the device, whatever order it traces in, must end up with exactly
    # those records.
    small = max(1, ref["kept"] // 3)
    first = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=1, sink_mode=O.SINK_PROBE,
                      probes=probes, probe_cell_size=keep, keep_radius=keep, store_capacity=small)
    assert first["store_size"] == small and first["store_dropped"] == ref["kept"] - small
    tracer.set_probes(probes, keep, keep, small)
    tracer.reset()
    tracer.trace(sink_mode=1)
    c = tracer.counters()
    assert c["store_size"] == small and c["store_dropped"] == ref["kept"] - small and c["store_attempted"] == ref["kept"]
    assert_bits_equal(_sorted_records(tracer.store()), _sorted_records(first["store"]), "records of the overflowing store")


@pytest.mark.parametrize("variant", ["one pass", "two passes", "bounce levels", "radix order", "preloaded"])
def test_overflowing_store_holds_the_first_records_in_photon_order(rt, scene_opener, variant, monkeypatch):
    """The headline configuration's regime: the probes keep several times the records the store has room
    for (MirrorBombardment-200M keeps 32 M for 7.5 M places).  The reference's store then holds the
    deposits of the photons its workers met first, i.e. of the lowest indices — a fair sample of the
    emission directions.  The device traces in emission order, run by run, so what ARRIVES first is
    anything but: it takes the overflow into room behind the capacity and settles it by photon order
    afterwards.  The oracle's single thread walks the photons in the reference's order; the stores must
    hold the same records bit for bit — also when the pass is split in two ranges, goes through the
    bounce-level queues or the radix-sorted schedule, or starts on a store that already holds records."""
    if variant == "bounce levels":
        monkeypatch.setenv("RT_AMD_QUEUE_MIN", "1")
        monkeypatch.setenv("RT_AMD_LEVEL_SORT_MIN", "1")
    if variant == "radix order":
        monkeypatch.setenv("RT_AMD_EMISSION_SORT", "radix")
    photons = 300_000
    name = "cbgmd" if variant == "bounce levels" else "bombardment"
    entry = dict(SCENES[name], photons_per_light=photons)
    scene, tracer, orc = _open(rt, scene_opener, entry)
    pre = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=8, record_photons=4000)
    probes = pre["bounce_f64"][:, 0, :3][pre["casts_valid"][:, 1] > 0][::2]
    extent = np.ptp(probes, axis=0).max()
    keep = float(extent) / 40.0
    whole = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=8, sink_mode=O.SINK_PROBE,
                      probes=probes, probe_cell_size=keep, keep_radius=keep, store_capacity=1 << 24)
    preload = whole["store"][:777].copy() if variant == "preloaded" else None
    capacity = max(1000, whole["kept"] // 4) + (777 if preload is not None else 0)
    assert whole["kept"] > 4000
    want = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=1, sink_mode=O.SINK_PROBE, probes=probes,
                     probe_cell_size=keep, keep_radius=keep, store_capacity=capacity, preload=preload)
    tracer.set_probes(probes, keep, keep, capacity)
    tracer.reset()
    if preload is not None:
        tracer.store_append(preload)
    if variant == "two passes":
        tracer.trace(sink_mode=1, first_photon=0, photon_count=photons // 5)
        tracer.trace(sink_mode=1, first_photon=photons // 5, photon_count=photons - photons // 5)
    else:
        tracer.trace(sink_mode=1)
    c = tracer.counters()
    for k in ("kept", "culled", "store_attempted", "store_size", "store_dropped"):
        assert c[k] == want[k], k
    assert c["store_size"] == capacity and c["store_dropped"] > 2 * capacity
    got = tracer.store()
    assert_bits_equal(_sorted_records(got), _sorted_records(want["store"]), "records of the overflowing store")
    if preload is not None:  # what was there before the pass stays where it was
        assert_bits_equal(got[:777], preload, "preloaded records")
    # and the survivors are spread over the whole emission domain, not bunched where the device happened
    # to start: the centroid of the kept records is the centroid of everything the probes keep
    everything = whole["store"][:whole["store_size"], :3].astype(np.float64)
    held = got[(777 if preload is not None else 0):, :3].astype(np.float64)
    spread = everything.std(axis=0).max()
    assert np.abs(held.mean(axis=0) - everything.mean(axis=0)).max() < 0.05 * spread


def test_store_overflowing_its_room_too_still_counts_right(rt, scene_opener, monkeypatch, capfd):
    """RT_AMD_STORE_ROOM holds the room behind the capacity down to a few records: the pass overflows
    that as well.  The counters stay exact, the store is full, and the library says on stderr that the
    survivors are no longer the first in photon order."""
    monkeypatch.setenv("RT_AMD_STORE_ROOM", "64")
    entry = dict(SCENES["bombardment"], photons_per_light=60_000)
    scene, tracer, orc = _open(rt, scene_opener, entry)
    pre = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=8, record_photons=2000)
    probes = pre["bounce_f64"][:, 0, :3][pre["casts_valid"][:, 1] > 0][::2]
    keep = float(np.ptp(probes, axis=0).max()) / 40.0
    want = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=1, sink_mode=O.SINK_PROBE, probes=probes,
                     probe_cell_size=keep, keep_radius=keep, store_capacity=500)
    assert want["store_dropped"] > 1000
    tracer.set_probes(probes, keep, keep, 500)
    tracer.reset()
    tracer.trace(sink_mode=1)
    store = tracer.store()
    c = tracer.counters()
    for k in ("kept", "store_attempted", "store_size", "store_dropped"):
        assert c[k] == want[k], k
    assert store.shape[0] == 500 and np.isfinite(store[:, :9]).all()
    assert "overflowed its room as well" in capfd.readouterr().err


def _philox_first_block(seed, light, photon):
    """Words 0 and 2 of Philox4x32-10 block 0 of the emission stage (include/rt_portable.h) for an array
    of photon indices: the leading bits of the first two emission draws."""
    m0, m1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    shift = np.uint64(32)
    c = [photon.astype(np.uint64) & mask, photon.astype(np.uint64) >> shift, np.zeros(photon.shape, np.uint64),
         np.zeros(photon.shape, np.uint64)]
    k0, k1 = np.uint64(seed), np.uint64(light)
    for _ in range(10):
        p0, p1 = m0 * c[0], m1 * c[2]
        c = [(p1 >> shift) ^ c[1] ^ k0, p1 & mask, (p0 >> shift) ^ c[3] ^ k1, p0 & mask]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & mask, (k1 + np.uint64(0xBB67AE85)) & mask
    return c[0], c[2]


def _spread12(v):
    v = v.astype(np.uint32)
    v = (v | (v << np.uint32(8))) & np.uint32(0x00FF00FF)
    v = (v | (v << np.uint32(4))) & np.uint32(0x0F0F0F0F)
    v = (v | (v << np.uint32(2))) & np.uint32(0x33333333)
    v = (v | (v << np.uint32(1))) & np.uint32(0x55555555)
    return v


@pytest.mark.parametrize("name,sort,photons,first", [("bombardment", "radix", 1_300_000, 0),
                                                     ("bombardment", "refine", 1_300_000, 12345),
                                                     ("bombardment", "refine", 3_001, 0),
                                                     ("bombardment", "default", 4_500_000, 0),
                                                     ("bombardment", "default", 1_300_000, 0),
                                                     ("cbgmd", "refine", 700_000, 3),
                                                     ("mixed", "refine", 200_000, 3)])
def test_emission_order_is_a_sorted_permutation(rt, scene_opener, name, sort, photons, first, monkeypatch):
    """The schedule of a pass: every id of the pass exactly once, ascending by emission cell (the Morton
    number of the cell of the first two emission draws in a 4096 x 4096 grid).  Large passes sort
    all but the key's low eight bits (passes of 1 Mi to 2 Mi ids: its leading eight) and settle the rest inside every run
    of 4096 positions: there the runs
    follow each other in the order of the cell without its low bits, and inside a run the cell ascends
    (the whole cell where the run spans few values of the leading bits, without some of its last bits
    where it spans many).  Checked for
    the radix sort over all bits, for the two-step order forced onto small passes (one of a range that
    does not start at photon 0, one shorter than a run) and for the order a large pass picks by itself;
    an area light's key interleaves four draws and several lights put the light above the cell: those
    are only checked to be permutations here."""
    if sort != "default":
        monkeypatch.setenv("RT_AMD_EMISSION_SORT", sort)
    entry = dict(SCENES[name], photons_per_light=photons + first)
    scene, tracer, _ = _open(rt, scene_opener, entry)
    tracer.reset()
    tracer.trace(sink_mode=0, first_photon=first, photon_count=photons)
    order = tracer.emission_order()
    total = photons * scene.desc.light_count
    assert order.size == total
    assert np.array_equal(np.sort(order), np.arange(total, dtype=np.uint32)), "a permutation of the pass"
    assert tracer.counters()["photons"] == total
    if name != "bombardment":
        return
    seed = int(scene.settings.seed) if scene.settings.seeded else 0
    w0, w2 = _philox_first_block(seed, 0, order.astype(np.uint64) + np.uint64(first))
    cell = (_spread12(w0 >> np.uint64(20)) | (_spread12(w2 >> np.uint64(20)) << np.uint32(1))).astype(np.int64)
    if sort == "radix":
        assert np.all(np.diff(cell) >= 0), "ascending by emission cell"
        return
    # a pass of 1 Mi to 2 Mi ids that picks its order by itself sorts the leading EIGHT bits before the runs are settled
    lead = 16 if sort == "default" and (1 << 20) <= total < (2 << 20) else 8
    last = -1
    for begin in range(0, total, 4096):
        run = cell[begin:begin + 4096]
        assert (run >> lead).min() >= last, f"run at {begin}: the runs ascend by the cell's leading bits"
        last = (run >> lead).max()
        # a run that spans many values of the leading bits drops low bits until 2048 bins hold it
        lowest, drop = (run[0] >> lead) << lead, 0
        while (((run[-1] | ((1 << lead) - 1)) - lowest) >> drop) >= 2048:
            drop += 1
        assert np.all(np.diff((run - lowest) >> drop) >= 0), f"run at {begin}: ascending by the cell"


@pytest.mark.parametrize("name,photons,sort", [("bombardment", 1_300_000, "default"), ("mixed", 250_000, "default"),
                                               ("bombardment", 1_300_000, "refine"), ("mixed", 250_000, "refine"),
                                               ("cbgmd", 1_100_000, "refine")])
def test_sorted_pass_equals_oracle(rt, scene_opener, name, photons, sort, monkeypatch):
    """Passes are traced in emission order (radix sort of the ids by emission cell, the light above
    it: the multi-light case interleaves lights in id order but not in the sorted order; large passes
    settle the key's low bits run by run after the sort, forced here onto small ones).
    The schedule must not change any result: same counters, same store multiset, same paths
    for the recorded photons (record slots follow photon ids, not the schedule)."""
    if sort != "default":
        monkeypatch.setenv("RT_AMD_EMISSION_SORT", sort)
    entry = dict(SCENES[name], photons_per_light=photons)
    scene, tracer, orc = _open(rt, scene_opener, entry)
    assert photons * scene.desc.light_count >= 1 << 20
    setup = O.frame_setup(orc, O.RNG_COUNTER, O.MATH_PORTABLE)
    keep = setup["keep_radius"]
    capacity = 1 << 22
    ref = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=8, sink_mode=O.SINK_PROBE,
                    probes=setup["probes"], probe_cell_size=keep, keep_radius=keep, store_capacity=capacity,
                    record_photons=5000)
    tracer.set_probes(setup["probes"], keep, keep, capacity)
    tracer.reset()
    tracer.trace(sink_mode=1, record_photons=5000)
    c = tracer.counters()
    for k in ("photons", "casts", "bounces", "deposits", "kept", "culled", "store_size"):
        assert c[k] == ref[k], k
    assert ref["kept"] > 1000
    assert_bits_equal(_sorted_records(tracer.store()), _sorted_records(ref["store"]), "store records")
    got = tracer.records()
    assert_bits_equal(got["emitted"], ref["emitted"], "emitted")
    assert np.array_equal(got["casts_valid"], ref["casts_valid"])
    assert_bits_equal(got["bounce_f64"], ref["bounce_f64"], "hits")
    assert np.array_equal(got["bounce_i32"], ref["bounce_i32"]), "ids"
    sort_ms, trace_ms = tracer.timings()
    assert sort_ms > 0.0 and trace_ms > 0.0
    # the frames of a sequence may reuse the order: same results, no sort; another range sorts anew
    tracer.reset()
    tracer.trace(sink_mode=1)
    sort_ms = tracer.timings()[0]  # with the work space in place
    tracer.reset()
    tracer.trace(sink_mode=1, record_photons=5000, reuse_order=True)
    again = tracer.counters()
    for k in ("photons", "casts", "bounces", "deposits", "kept", "culled", "store_size"):
        assert again[k] == ref[k], k
    assert_bits_equal(_sorted_records(tracer.store()), _sorted_records(ref["store"]), "store records, order reused")
    assert_bits_equal(tracer.records()["bounce_f64"], ref["bounce_f64"], "hits, order reused")
    assert tracer.timings()[0] < 0.5 * sort_ms
    tracer.reset()
    tracer.trace(sink_mode=1, first_photon=7, photon_count=photons - 7, reuse_order=True)
    assert tracer.counters()["photons"] == (photons - 7) * scene.desc.light_count and tracer.timings()[0] > 0.5 * sort_ms


PROBE_SCENES = {
    "mixed": {"scene": "tests/scenes/mixed.json"},
    "bombardment": {"scene": "scenes/MirrorBombardment-200M.json", "image": 160, "seed": 11},
    "cbgmd": {"scene": "scenes/CornellBoxGlassMirrorDiffuse.json", "image": 128, "probe_sub_sample": 2, "seed": 3},
    "multicam": {"scene": "scenes/MultiCameraTest.json", "image": 96, "seed": 5},
    "shutter": {"scene": "tests/scenes/mixed.json", "shutter_time": 0.25, "camera_time_samples": 4},
    "animated": {"scene": "tests/scenes/animated.json", "frame_time": 0.3},
    "stochastic128": {"scene": "scenes/CBGMD_stochastic128.json", "seed": 4},
    # thin-lens cameras (aperture given, aperture from focal length and f-number) and an orthographic one
    "lens": {"scene": "tests/scenes/lens.json"},
}


@pytest.mark.parametrize("name", sorted(PROBE_SCENES))
def test_probe_pass_equals_oracle(rt, scene_opener, name):
    """The device probe pass (one thread per pixel, count + write launches) returns the
    oracle's gather points bit for bit, in the same pixel-major order, with the same counters."""
    scene, tracer, orc = _open(rt, scene_opener, PROBE_SCENES[name])
    total = 0
    for c in range(scene.desc.camera_count):
        got, stats = tracer.collect_gather_points(c)
        want, ostats = orc.collect_gather_points(c, rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE)
        assert stats == ostats
        assert_bits_equal(got, want, f"{name} camera {c}")
        total += got.shape[0]
    assert total > 0
    assert_bits_equal(tracer.collect_probe_positions(), orc.collect_probe_positions(), "probe union")


def test_probe_pass_equals_reference_where_rng_free(rt, golden, manifest, scene_opener):
    """Pixels that draw no random numbers (no glass, no lens, no shutter) must equal the
    REFERENCE's records bit for bit; on the bombardment scene that is every pixel whose
    first hit is not the glass sphere."""
    g = golden("probes_bombardment")
    scene, tracer, _ = _open(rt, scene_opener, manifest["probes_bombardment"])
    got, stats = tracer.collect_gather_points(0)
    want = g["probes_c0"]
    assert stats["camera_rays"] == int(g["probes_counts_c0"][0])
    direct_got, direct_want = got[got[:, 18] == 1.0], want[want[:, 18] == 1.0]
    assert direct_want.shape[0] > 400
    # sin/cos/acos play no part in pinhole rays or hits: libm vs portable math is moot here
    assert_bits_equal(direct_got, direct_want, "RNG-free gather points vs reference")


def test_probe_pass_in_one_pass_of_casts_equals_count_then_write(rt, scene_opener, monkeypatch):
    """A camera whose pixels yield one record at most (no lens or shutter samples, no dielectric in the scene)
    is collected in one pass of casts — a lean launch for every pixel's camera ray, a second one for the pixels
    that see a mirror, records left at their pixels until somebody asks for them — and otherwise by counting
    and then writing: the same records to the bit, the same counts, and the same again after
    rt_forget_gather_points; the set-up's timings are those of the run that was made."""
    entry = {"scene": "scenes/MirrorBombardment-200M.json", "photons_per_light": 20000, "image": 160, "seed": 11}
    scene, tracer, orc = _open(rt, scene_opener, entry)
    setup = O.frame_setup(orc, O.RNG_COUNTER, O.MATH_PORTABLE)
    keep = setup["keep_radius"]
    points, stats = tracer.collect_gather_points(0)
    assert len(points) > 1000 and stats["delta_extensions"] > 0, "some pixels see the mirror"
    probes, capacity = tracer.set_probes_from_cameras(keep, keep, 256, int(scene.settings.bounce_store_capacity))
    assert probes == len(points)
    timings = tracer.setup_timings()
    assert timings["probe_pass"] > 0 and timings["index_build"] > 0 and timings["filter_build"] > 0
    tracer.reset()
    tracer.trace(sink_mode=1)
    tracer.gather(0, setup["gather_cell"], setup["min_gather_radius"])
    counters, store = tracer.counters(), _sorted_records(tracer.store())
    buffer = tracer.camera_buffer(0).copy()

    tracer.forget_gather_points()
    again, stats_again = tracer.collect_gather_points(0)
    assert_bits_equal(again, points, "the pass run again")
    assert stats_again == stats

    monkeypatch.setenv("RT_AMD_PROBE_PASS", "two")
    other = rt.Tracer(scene)
    two, stats_two = other.collect_gather_points(0)
    assert_bits_equal(two, points, "count then write")
    assert stats_two == stats
    assert other.set_probes_from_cameras(keep, keep, 256, int(scene.settings.bounce_store_capacity)) == (probes, capacity)
    other.reset()
    other.trace(sink_mode=1)
    other.gather(0, setup["gather_cell"], setup["min_gather_radius"])
    for k in ("photons", "casts", "bounces", "deposits", "kept", "culled", "store_size"):
        assert other.counters()[k] == counters[k], k
    assert_bits_equal(_sorted_records(other.store()), store, "store records")
    assert_bits_equal(other.camera_buffer(0), buffer, "gathered frame")


@pytest.mark.parametrize("name", ["mixed", "multicam", "animated"])
def test_probes_from_cameras_equal_probes_through_the_host(rt, scene_opener, name):
    """rt_set_probes_from_cameras builds the keep-test index from the probe-pass records on the
    device; the photon pass then keeps exactly what it keeps with the same positions handed over
    by the caller, and the store is sized by the reference's rule."""
    entry = dict(PROBE_SCENES[name], photons_per_light=4000)
    scene, tracer, orc = _open(rt, scene_opener, entry)
    setup = O.frame_setup(orc, O.RNG_COUNTER, O.MATH_PORTABLE)
    keep = setup["keep_radius"]
    probes = tracer.collect_probe_positions()
    tracer.set_probes(probes, keep, keep, setup["store_capacity"])
    tracer.reset()
    tracer.trace(sink_mode=1)
    want_counters, want_store = tracer.counters(), _sorted_records(tracer.store())

    other = rt.Tracer(scene)
    count, capacity = other.set_probes_from_cameras(keep, keep, 256, int(scene.settings.bounce_store_capacity))
    assert count == probes.shape[0] and capacity == setup["store_capacity"]
    other.reset()
    other.trace(sink_mode=1)
    got = other.counters()
    for k in ("kept", "culled", "store_size", "deposits"):
        assert got[k] == want_counters[k], k
    assert want_counters["kept"] > 0
    assert_bits_equal(_sorted_records(other.store()), want_store, "store records")


@pytest.mark.parametrize("name", ["mixed", "cbgmd", "animated"])
def test_emitter_deposits_equal_oracle(rt, scene_opener, name):
    """rt_deposit_emitters tiles the lit faces on the device: the same records as the oracle's
    ProbeGather::depositEmitters (as a multiset: the device appends in no particular order)."""
    scene, tracer, orc = _open(rt, scene_opener, GATHER_SCENES[name])
    setup = O.frame_setup(orc, O.RNG_COUNTER, O.MATH_PORTABLE)
    keep = setup["keep_radius"]
    want, stats = orc.deposit_emitters(setup["probes"], keep, keep, setup["emitter_spacing"])
    tracer.set_probes_from_cameras(keep, keep, 256, int(scene.settings.bounce_store_capacity))
    tracer.reset()
    samples, kept = tracer.deposit_emitters(setup["emitter_spacing"])
    assert kept == len(want) and kept > 0 and samples == stats["generated"]
    assert tracer.counters()["store_size"] == kept
    assert_bits_equal(_sorted_records(tracer.store()), _sorted_records(np.asarray(want)), "emitter deposits")
    # a second frame over the same probes gets the records the library kept from the first ...
    tracer.reset()
    assert tracer.deposit_emitters(setup["emitter_spacing"]) == (samples, kept)
    assert tracer.counters()["store_size"] == kept
    assert_bits_equal(_sorted_records(tracer.store()), _sorted_records(np.asarray(want)), "emitter deposits, kept copy")
    # ... and deposits made into a store that already holds records are worked out anew and appended
    assert tracer.deposit_emitters(setup["emitter_spacing"]) == (samples, kept)
    assert tracer.counters()["store_size"] == 2 * kept
    # new probes: nothing of the old run is reused
    tracer.set_probes(setup["probes"][: max(1, len(setup["probes"]) // 2)], keep, keep, int(scene.settings.bounce_store_capacity))
    tracer.reset()
    half, _ = orc.deposit_emitters(setup["probes"][: max(1, len(setup["probes"]) // 2)], keep, keep, setup["emitter_spacing"])
    assert tracer.deposit_emitters(setup["emitter_spacing"])[1] == len(half)


def test_probe_keep_test_matches_store_decisions(rt, scene_opener):
    """rt_probe_keep_test gives, for points handed over from the host, the same decision the
    photon pass takes for its own deposits."""
    scene, tracer, orc = _open(rt, scene_opener, SCENES["mixed"])
    pre = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=8, record_photons=3000)
    valid = pre["casts_valid"][:, 1] > 0
    points = pre["bounce_f64"][:, 0, :3][valid]
    probes = points[::7]
    keep = 9.0
    tracer.set_probes(probes, keep, keep, 1024)
    got = tracer.probe_keep_test(points)
    d2 = ((points[:, None, :] - probes[None, :, :]) ** 2).sum(-1).min(1)
    clear = np.abs(np.sqrt(d2) - keep) > 1e-9
    assert np.array_equal(got[clear], (d2 <= keep * keep)[clear])
    assert got[::7].all()
    # a cell size below the keep radius (reach 2) takes the unfiltered scan
    tracer.set_probes(probes, 5.0, keep, 1024)
    assert np.array_equal(tracer.probe_keep_test(points), got)


@pytest.mark.parametrize("cell_over_keep", [1.0, 2.5])
def test_keep_test_equals_the_scan_over_all_probes(rt, scene_opener, cell_over_keep):
    """The keep test as the device index answers it (bricks of cells over a dense cell array, the cells
    around a point picked by their distance from it) against the definition, reference
    src/ProbeIndex.cpp:27-68: some probe lies within the keep radius, by the reference's expression in the
    reference's operation order.  Probes on both sides of the origin, on cell borders, crowded into single
    cells (walked several a trip), alone in their bricks; points at the probes, a keep radius away from them
    to the last bit or two, across cell and brick borders, and far from every probe."""
    scene, tracer, orc = _open(rt, scene_opener, SCENES["mixed"])
    rng = np.random.default_rng(17)
    keep = 0.75
    cell = keep * cell_over_keep
    sheet = np.stack([rng.uniform(-20, 20, 6000), rng.uniform(-20, 20, 6000), np.zeros(6000)], axis=1)
    sheet[:, 2] = 0.3 * np.sin(sheet[:, 0]) + 0.05 * sheet[:, 1]          # a surface through many bricks
    on_borders = np.round(rng.uniform(-8, 8, (500, 3)) / cell) * cell       # coordinates that are multiples of the cell
    crowded = np.array([3.1, -2.2, 7.7]) * cell + rng.uniform(0, 1, (70, 3)) * cell * 0.999  # seventy probes in one cell
    crowded2 = np.array([-5.0, 4.0, -6.0]) * cell + rng.uniform(0, 1, (5, 3)) * cell * 0.999
    lonely = rng.uniform(-300, 300, (40, 3))
    probes = np.concatenate([sheet, on_borders, crowded, crowded2, lonely])
    rng.shuffle(probes)
    tracer.set_probes(probes, cell, keep, 4096)

    directions = rng.normal(size=(4000, 3))
    directions /= np.linalg.norm(directions, axis=1)[:, None]
    picked = probes[rng.integers(0, len(probes), 4000)]
    points = np.concatenate([
        probes[:3000],                                            # at a probe
        picked + directions * keep,                               # a keep radius away, as closely as rounding lets
        picked + directions * keep * (1 + 1e-15),
        picked + directions * keep * (1 - 1e-15),
        picked + directions * rng.uniform(0, 2.5, (4000, 1)) * keep,
        crowded.mean(0) + directions[:500] * rng.uniform(0.5, 1.5, (500, 1)) * keep,
        rng.uniform(-25, 25, (4000, 3)),
        rng.uniform(-400, 400, (500, 3)),
        np.array([[1e7, 0.0, 0.0], [-1e7, 3.0, 2.0], [0.0, 0.0, 2e6 * cell]]),   # beyond the cells that pack
    ])
    got = tracer.probe_keep_test(points)
    want = np.zeros(len(points), dtype=bool)
    for start in range(0, len(points), 512):
        d = probes[None, :, :] - points[start:start + 512, None, :]
        d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]
        want[start:start + 512] = (d2 <= keep * keep).any(1)
    assert want[:3000].all() and 0.2 < want.mean() < 0.9
    assert np.array_equal(got.astype(bool), want), np.nonzero(got.astype(bool) != want)[0][:10]
    # the smallest grids: one probe, and a handful in one cell
    for few in (probes[:1], crowded2):
        tracer.set_probes(few, cell, keep, 64)
        near = few[0] + directions[:200] * rng.uniform(0, 2.0, (200, 1)) * keep
        d = few[None, :, :] - near[:, None, :]
        d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]
        assert np.array_equal(tracer.probe_keep_test(near).astype(bool), (d2 <= keep * keep).any(1))


GATHER_SCENES = {
    "mixed": {"scene": "tests/scenes/mixed.json", "photons_per_light": 2500},
    "mixed_tight": {"scene": "tests/scenes/mixed.json", "photons_per_light": 2500, "probe_keep_radius_scale": 2.0},
    "cbgmd": {"scene": "scenes/CornellBoxGlassMirrorDiffuse.json", "photons_per_light": 60000, "image": 96, "seed": 3},
    "bombardment": {"scene": "scenes/MirrorBombardment-200M.json", "photons_per_light": 400000, "image": 128,
                    "seed": 11},
    "animated": {"scene": "tests/scenes/animated.json", "photons_per_light": 2500, "frame_time": 0.3},
    "stochastic128": {"scene": "scenes/CBGMD_stochastic128.json", "photons_per_light": 60000, "seed": 4},
    "lens": {"scene": "tests/scenes/lens.json", "photons_per_light": 1500},
}


@pytest.mark.parametrize("lanes,hand_over", [(0, None), (1, 0), (1, 3), (8, None), (64, None), (-1, None), (-4, None), (-8, None)])
@pytest.mark.parametrize("name", sorted(GATHER_SCENES))
def test_gather_equals_oracle(rt, scene_opener, name, lanes, hand_over, monkeypatch):
    """(The gather gives a sampled pixel a lane and hands it to a wave when it meets more records than a
    limit: (0, None) is its own choice — the pixels handed over go to two waves each; -1, -4 and -8 make that one,
    four or eight waves —, (1, 0) keeps every pixel in its lane, (1, 3) mixes both, 8 and 64 force
    that many lanes per pixel in one launch; the result may not depend on it by a single bit.)
    The whole probe-mode frame after the photon pass.  (1) Given the SAME store, record for
    record in the same order, the device gather (sort by cell, one thread per pixel) returns
    the oracle's camera buffers bit for bit.  (2) The device's own frame (emitter deposits
    pre-loaded, photon pass, gather) holds the same store as a multiset and the same image up
    to the rounding of a different float summation order (the reference's own result depends
    on append order the same way when it runs more than one worker)."""
    if lanes < 0:
        # its own choice, but the pixels handed over are taken by 1, 4 or 8 waves each instead of by two
        monkeypatch.setenv("RT_AMD_GATHER_WAVES", str(-lanes))
    elif lanes:
        monkeypatch.setenv("RT_AMD_GATHER_LANES", str(lanes))
    if hand_over is not None:
        monkeypatch.setenv("RT_AMD_GATHER_HEAVY", str(hand_over))
    scene, tracer, orc = _open(rt, scene_opener, GATHER_SCENES[name])
    setup = O.frame_setup(orc, O.RNG_COUNTER, O.MATH_PORTABLE)
    keep = setup["keep_radius"]
    emitters, _ = orc.deposit_emitters(setup["probes"], keep, keep, setup["emitter_spacing"])
    ref = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, sink_mode=O.SINK_PROBE, threads=1,
                    probes=setup["probes"], probe_cell_size=keep, keep_radius=keep,
                    store_capacity=setup["store_capacity"], preload=emitters)
    assert ref["store_size"] > len(emitters)
    cameras = [c for c in range(scene.desc.camera_count)
               if scene.desc.cameras[c].bounce_filter < 0 and scene.desc.cameras[c].light_filter < 0]
    want = {c: orc.gather(c, ref["store"], setup["gather_cell"], setup["min_gather_radius"]) for c in cameras}

    tracer.set_probes(tracer.collect_probe_positions(), keep, keep, setup["store_capacity"])
    # (1) same store, same order — with every form of index over the sorted store: the directory of
    # cells (what these small scenes get by themselves), a directory of rows of cells (large scenes:
    # the rows are searched for the cells wanted) or none at all (the whole list is searched)
    indexes = {"cells": {}, "rows": {"RT_AMD_GATHER_DIRECTORY": "3000"}, "none": {"RT_AMD_GATHER_DIRECTORY": "1"}}
    for index, env in indexes.items():
        monkeypatch.delenv("RT_AMD_GATHER_DIRECTORY", raising=False)
        for key, value in env.items():
            monkeypatch.setenv(key, value)
        tracer.reset()
        tracer.store_append(ref["store"])
        for c in cameras:
            stats = tracer.gather(c, setup["gather_cell"], setup["min_gather_radius"])
            buffer, ostats = want[c]
            assert stats["unindexed"] == 0
            assert (stats["pixels_hit"], stats["pixels_gathered"], stats["deposits_accum"]) == (
                ostats["pixels_hit"], ostats["pixels_gathered"], ostats["deposits_accum"]), index
            assert ostats["pixels_gathered"] > 0
            assert_bits_equal(tracer.camera_buffer(c), buffer, f"{name} camera {c} gathered buffer, index: {index}")
    monkeypatch.delenv("RT_AMD_GATHER_DIRECTORY", raising=False)
    # (2) the device's own frame
    tracer.reset()
    tracer.store_append(emitters)
    tracer.trace(sink_mode=1)
    counters = tracer.counters()
    assert counters["store_size"] == ref["store_size"] and counters["kept"] == ref["kept"]
    assert_bits_equal(_sorted_records(tracer.store()), _sorted_records(ref["store"]), "store records")
    for c in cameras:
        stats = tracer.gather(c, setup["gather_cell"], setup["min_gather_radius"])
        buffer, ostats = want[c]
        assert stats["deposits_accum"] == ostats["deposits_accum"]
        got = tracer.camera_buffer(c)
        np.testing.assert_allclose(got, buffer, rtol=2e-4, atol=1e-6 * float(np.abs(buffer).max()))


def test_knot4k_frame_at_its_own_size(rt, scene_opener):
    """MirrorKnot4k at its own 4096 x 4096 (16.7 M probes, 2 GB of keep-test tables, a 40 M-record store) with a
    quarter of its photons: probe pass, emitter deposits, photon pass and gather on the device, checked
    against the oracle on a 256 x 256 crop.  The oracle visits the crop's pixels only (every pixel has its
    own counter stream), keeps what lies within the keep radius of the crop's probes — a subset of what the
    device keeps with all probes, and a superset of what the crop's gather points can reach, since the keep
    radius exceeds every gather radius — and gathers it: the device's store must contain every such record
    bit for bit, and its buffer must equal the oracle's on the crop up to float summation order (the
    device appends in no particular order)."""
    x0, y0, side = 1920, 2300, 256
    entry = {"scene": "scenes/MirrorKnot4k.json", "photons_per_light": 5_000_000}
    scene, tracer, orc = _open(rt, scene_opener, entry)
    assert scene.desc.cameras[0].width == 4096 and scene.desc.cameras[0].height == 4096
    orc.set_pixel_window(x0, y0, x0 + side, y0 + side)
    setup = O.frame_setup(orc, O.RNG_COUNTER, O.MATH_PORTABLE)  # the crop's probes
    keep = setup["keep_radius"]
    rows, _ = orc.collect_gather_points(0)
    assert rows.shape[0] >= side * side // 2
    assert keep >= max(float(rows[:, 16].max()), setup["min_gather_radius"]), "keep radius must cover every gather radius"
    emitters, _ = orc.deposit_emitters(setup["probes"], keep, keep, setup["emitter_spacing"])
    capacity = int(scene.settings.bounce_store_capacity)
    ref = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, sink_mode=O.SINK_PROBE, threads=16,
                    probes=setup["probes"], probe_cell_size=keep, keep_radius=keep, store_capacity=capacity,
                    preload=emitters)
    want, ostats = orc.gather(0, ref["store"], setup["gather_cell"], setup["min_gather_radius"])
    assert ostats["pixels_gathered"] > 200

    probes, store_capacity = tracer.set_probes_from_cameras(keep, keep, 256, capacity)
    assert probes > 16_000_000 and store_capacity == capacity
    tracer.reset()
    tracer.deposit_emitters(setup["emitter_spacing"])
    tracer.trace(sink_mode=1)
    c = tracer.counters()
    assert c["photons"] == 5_000_000 and c["bounces"] == ref["bounces"] and c["deposits"] == ref["deposits"]
    assert c["store_dropped"] == 0 and c["store_size"] >= ref["store_size"]
    store = tracer.store()
    row = np.dtype((np.void, 52))
    device_rows = np.ascontiguousarray(store).view(row).ravel()
    oracle_rows = np.ascontiguousarray(ref["store"]).view(row).ravel()
    assert np.isin(oracle_rows, device_rows).all(), "a record the crop's probes keep is missing from the device's store"
    stats = tracer.gather(0, setup["gather_cell"], setup["min_gather_radius"])
    assert stats["unindexed"] == 0 and stats["pixels_hit"] == probes
    got = tracer.camera_buffer(0)[y0:y0 + side, x0:x0 + side]
    expected = want[y0:y0 + side, x0:x0 + side]
    assert np.array_equal(got > 0, expected > 0), "lit pixels of the crop differ"
    np.testing.assert_allclose(got, expected, rtol=2e-4, atol=1e-6 * float(np.abs(expected).max()))


@pytest.mark.parametrize("key", ["render_mixed", "render_animated"])
def test_frame_statistically_equal_to_reference_frame(rt, golden, manifest, scene_opener, key):
    """Image equivalence at equal photon count against the REFERENCE's own frame.  The device draws
    from Philox, the reference from one mt19937 per worker, so the two frames are two samples of
    the same estimator: their relative RMSE must be what two reference-equivalent runs with
    different seeds give (the oracle in mt19937 / libm mode, which reproduces the reference's frame
    bit for bit with the reference's seed).  Stated tolerance: relRMSE(device, reference) <=
    1.25 x the largest relRMSE among three such runs and the reference, and total energy within 5%
    of the mean of those four frames.  (Over twelve seeds the frame energy of these scenes has a
    standard deviation of 1.3% in both generator modes, and the two modes' means agree to 0.5%,
    inside their standard error; 5% is four standard deviations.)"""
    image_check = __import__("importlib").import_module("ray-tracer_amd.image_check")
    g = golden(key)
    reference = g["render_buffer_c0"].astype(np.float64)
    entry = dict(manifest[key])
    samples = []
    for seed in (101, 202, 303):
        scene = scene_opener(dict(entry, seed=seed))
        orc = O.Oracle(scene)
        setup = O.frame_setup(orc, O.RNG_MT, O.MATH_LIBM)
        keep = setup["keep_radius"]
        emitters, _ = orc.deposit_emitters(setup["probes"], keep, keep, setup["emitter_spacing"])
        r = orc.trace(rng_mode=O.RNG_MT, math_mode=O.MATH_LIBM, sink_mode=O.SINK_PROBE, probes=setup["probes"],
                      probe_cell_size=keep, keep_radius=keep, store_capacity=setup["store_capacity"], preload=emitters)
        samples.append(orc.gather(0, r["store"], setup["gather_cell"], setup["min_gather_radius"], O.RNG_MT,
                                  O.MATH_LIBM)[0].astype(np.float64))
    spread = max(image_check.rel_rmse(a, reference) for a in samples)
    spread = max([spread] + [image_check.rel_rmse(samples[i], samples[j]) for i in range(3) for j in range(i)])
    energies = [float(a.sum()) for a in samples] + [float(reference.sum())]

    scene, tracer, orc = _open(rt, scene_opener, entry)
    setup = O.frame_setup(orc, O.RNG_COUNTER, O.MATH_PORTABLE)
    keep = setup["keep_radius"]
    tracer.set_probes_from_cameras(keep, keep, 256, int(scene.settings.bounce_store_capacity))
    tracer.reset()
    tracer.deposit_emitters(setup["emitter_spacing"])
    tracer.trace(sink_mode=1)
    tracer.gather(0, setup["gather_cell"], setup["min_gather_radius"])
    device = tracer.camera_buffer(0).astype(np.float64)
    assert reference.sum() > 0 and spread > 0
    assert image_check.rel_rmse(device, reference) <= 1.25 * spread
    assert abs(float(device.sum()) - np.mean(energies)) <= 0.05 * np.mean(energies)


def _device_frame(rt, scene, setup, ranks, monkeypatch=None):
    """The probe-mode frame of `scene` on this GPU: `ranks` tracers, each with its share of the photons and
    of the frame's one store (rt_set_store_limit), emitter deposits on the first; the ranks' buffers added."""
    keep = setup["keep_radius"]
    photons = int(scene.settings.photons_per_light)
    total, counters = None, []
    for rank in range(ranks):
        tracer = rt.Tracer(scene)
        _, capacity = tracer.set_probes_from_cameras(keep, keep, 256, int(scene.settings.bounce_store_capacity))
        tracer.reset()
        _, emitters = tracer.deposit_emitters(setup["emitter_spacing"])
        if ranks > 1:
            tracer.reset()
            rest = capacity - min(emitters, capacity)
            tracer.set_store_limit(max(1, (min(emitters, capacity) if rank == 0 else 0) + (rest + ranks - 1) // ranks))
            if rank == 0:
                tracer.deposit_emitters(setup["emitter_spacing"])
        first, last = photons * rank // ranks, photons * (rank + 1) // ranks
        tracer.trace(sink_mode=1, first_photon=first, photon_count=last - first)
        tracer.gather(0, setup["gather_cell"], setup["min_gather_radius"])
        frame = tracer.camera_buffer(0).astype(np.float64)
        total = frame if total is None else total + frame
        counters.append(tracer.counters())
        tracer.close()
    return total, counters


@pytest.mark.parametrize("variant", ["one rank", "two ranks", "radix order", "bounce levels"])
def test_overflowing_headline_frame_against_the_reference_frame(rt, golden, manifest, scene_opener, variant, monkeypatch):
    """MirrorBombardment-200M runs with a store that overflows 4.3 times (32.1 M deposits kept by the probes
    for 7.5 M places), so its image is decided by WHICH deposits the store holds.  The fixture is the
    reference's own frame (Renderer::renderFrame, ten workers) of that scene at 160 x 160 with 4 M photons
    and a store of 150 000 records — the same 4.3-fold overflow — plus three frames of other seeds.
    The device's frame, on one rank and on two ranks that share the store, through the radix-sorted
    schedule and through the bounce-level queues, has to be one more such frame:
      * relative RMSE against each reference frame <= 1.25 x the largest among the reference frames,
      * total energy within four times their spread around their mean (the store holds exactly its
        capacity on both sides, so the energy hardly varies: 0.05 % between seeds),
      * on a 4 x 4 grid over the lit part of the image, every cell's mean within four times the
        reference frames' largest deviation from their mean in that cell — a schedule that filled the
        store from one side of the light's cone would show here.
    Reference: src/BounceStore.cpp:11-33 (appends beyond the capacity are dropped),
    src/Worker.cpp:429-505 (workers take the photons in index order), src/Renderer.cpp:382-390,653-667."""
    image_check = __import__("importlib").import_module("ray-tracer_amd.image_check")
    key = "render_bombardment_overflow"
    g, entry = golden(key), dict(manifest[key])
    references = [g["render_buffer_c0"].astype(np.float64)] + [g[f"spread{i}_buffer_c0"].astype(np.float64) for i in (1, 2, 3)]
    assert g["render_counters"][4] == entry["bounce_store_capacity"]          # the reference's store is full ...
    assert g["render_counters"][3] > 4.2 * entry["bounce_store_capacity"]    # ... 4.3 times over
    if variant == "radix order":
        monkeypatch.setenv("RT_AMD_EMISSION_SORT", "radix")
    if variant == "bounce levels":
        monkeypatch.setenv("RT_AMD_QUEUE_MIN", "1")
    scene, _, orc = _open(rt, scene_opener, entry)
    setup = O.frame_setup(orc, O.RNG_COUNTER, O.MATH_PORTABLE)
    device, counters = _device_frame(rt, scene, setup, 2 if variant == "two ranks" else 1)
    attempted = sum(c["store_attempted"] for c in counters)
    held = sum(c["store_size"] for c in counters)
    assert abs(attempted / g["render_counters"][3] - 1.0) < 0.01   # as many deposits kept by the probes
    assert entry["bounce_store_capacity"] <= held <= entry["bounce_store_capacity"] + 1

    assert_one_more_frame(device, references)


@pytest.mark.parametrize("name", ["mixed", "multicam", "cbgmd", "animated", "stochastic128"])
def test_splat_sink_equals_oracle(rt, scene_opener, name):
    """Sink B: direct camera splat (+ density grid).  Buffers are float-atomic sums, so they
    are compared to 1e-5 of the image's peak; counters are exact."""
    entry = dict(SCENES[name])
    scene, tracer, orc = _open(rt, scene_opener, entry)
    d = orc.derive()
    ref = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=8, sink_mode=O.SINK_SPLAT,
                    min_splat_radius=d["min_splat_radius"], density_cell_size=d["cell_size"])
    tracer.set_splat(d["min_splat_radius"], d["cell_size"], 1 << 18)
    tracer.reset()
    tracer.trace(sink_mode=2)
    c = tracer.counters()
    for k in ("splat_total", "splat_radius_clamped", "splat_luminance_clamped", "deposits", "density_cells",
              "density_deposits"):
        assert c[k] == ref[k], k
    assert ref["splat_total"] > 0 and c["density_dropped"] == 0
    for cam, expected in enumerate(ref["splat"]):
        got = tracer.camera_buffer(cam)
        assert np.array_equal(got > 0, expected > 0), f"camera {cam}: lit pixel sets differ"
        assert np.abs(got - expected).max() <= 1e-5 * max(expected.max(), 1e-30), f"camera {cam}"
    keys, power, counts = tracer.density()
    assert keys.shape[0] == ref["density_cells"] and int(counts.sum()) == ref["density_deposits"]
    assert np.allclose(power.astype(np.float64).sum(0), ref["deposited_power"], rtol=1e-4)


@pytest.mark.parametrize("name", ["MultiCameraTest", "CBGMD_stochastic128", "MirrorKnot4k", "CornellBoxGlassMirrorDiffuse"])
def test_density_grid_drops_nothing_at_config_budgets(rt, name):
    """The reference's DensityGrid grows without bound (src/DensityGrid.cpp:25-34); the device's table is
    sized from the photon budget when the sink is set up.  At every configuration's own budget the legacy
    sink must take every deposit: nothing dropped, as many grid deposits as non-delta bounces."""
    frame = __import__("importlib").import_module("ray-tracer_amd.frame")
    scene = rt.Scene(os.path.join(REPO, "scenes", name + ".json"))
    tracer = rt.Tracer(scene)
    frame.prepare_sinks(rt, scene, tracer, "splat")
    tracer.reset()
    tracer.trace(sink_mode=2)
    c = tracer.counters()
    assert c["photons"] == int(scene.settings.photons_per_light) * scene.desc.light_count
    assert c["density_dropped"] == 0 and c["density_deposits"] == c["deposits"] > 0
    assert c["splat_total"] > 0


def test_headline_pass_at_its_own_size(rt):
    """MirrorBombardment-200M with its 200 M photons and its probe sink, as the bench runs it.  What can be
    said at this size without an oracle run of 200 M photons: the schedule is a permutation of the pass,
    the counters obey the sink's bookkeeping, the pass traced as two halves adds up to the pass traced
    whole (every photon's path depends on its index alone), and the photons whose paths are recorded
    (the first 20 000 ids, traced wherever the schedule puts them) equal the oracle's bit for bit."""
    frame = __import__("importlib").import_module("ray-tracer_amd.frame")
    scene = rt.Scene(os.path.join(REPO, "scenes", "MirrorBombardment-200M.json"))
    photons = int(scene.settings.photons_per_light)
    assert photons == 200_000_000 and scene.desc.light_count == 1
    tracer = rt.Tracer(scene)
    frame.prepare_sinks(rt, scene, tracer, "probe")
    recorded = 20_000
    tracer.reset()
    tracer.trace(sink_mode=1, record_photons=recorded)
    whole, got = tracer.counters(), tracer.records()
    order = tracer.emission_order()
    assert order.size == photons
    seen = np.zeros(photons, dtype=bool)
    seen[order] = True
    assert seen.all(), "every id of the pass is scheduled (and, there being as many places as ids, once)"
    del seen, order
    assert whole["photons"] == photons and photons <= whole["casts"] and whole["bounces"] <= whole["casts"]
    assert whole["kept"] + whole["culled"] == whole["deposits"] and whole["store_attempted"] == whole["kept"]
    assert whole["store_size"] + whole["store_dropped"] == whole["store_attempted"]
    assert whole["bounces"] > photons // 2
    ref = O.Oracle(scene).trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=8, first_photon=0,
                                photon_count=recorded, record_photons=recorded)
    assert_bits_equal(got["emitted"], ref["emitted"], "emitted")
    assert np.array_equal(got["casts_valid"], ref["casts_valid"]), "casts / bounce counts"
    assert_bits_equal(got["bounce_f64"], ref["bounce_f64"], "hits")
    assert np.array_equal(got["bounce_i32"], ref["bounce_i32"]), "ids"
    half = photons // 2 + 12_345
    tracer.reset()
    tracer.trace(sink_mode=1, first_photon=0, photon_count=half)
    tracer.trace(sink_mode=1, first_photon=half, photon_count=photons - half)
    halves = tracer.counters()
    for k in ("photons", "casts", "bounces", "deposits", "kept", "culled", "store_attempted", "store_size", "store_dropped"):
        assert halves[k] == whole[k], k
    assert np.allclose(halves["deposited_power"], whole["deposited_power"], rtol=1e-9)


def test_furnace_ledger_at_scale(rt, scene_opener):
    """tests/test_FurnaceLedger.cpp:35-70 at N = 2 M on the GPU: the deposited power equals
    the closed form Phi (1 - rho^(B+1)) / (1 - rho) and every photon deposits exactly B+1
    times; photon sub-ranges add up (the multi-GPU partition, tests/test_CountEquivalence.cpp:85-135)."""
    n = 2_000_000
    scene, tracer, orc = _open(rt, scene_opener, {"scene": "tests/scenes/furnace.json", "photons_per_light": n})
    tracer.set_probes(np.zeros((1, 3)), 1e5, 1e5, 8)  # tiny store: everything beyond 8 records is "dropped"
    tracer.reset()
    tracer.trace(sink_mode=1)
    c = tracer.counters()
    terms = scene.settings.bounce_threshold + 1
    expected = 100.0 * 4 * np.pi * (1 - 0.5 ** terms) / 0.5
    # a handful of photons in ten million leak through a box edge (rounding of the quad
    # parameter test); the oracle leaks the same ones, so the counts are compared exactly
    # with it and to 1e-5 with the ideal N (B+1)
    ref = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=8)
    assert c["photons"] == n and c["bounces"] == ref["bounces"] and c["casts"] == ref["casts"]
    assert c["bounces"] == c["deposits"] == c["kept"] == c["store_attempted"]
    assert abs(c["bounces"] - n * terms) <= 1e-5 * n * terms
    assert c["culled"] == 0 and c["store_size"] == 8
    assert abs(c["deposited_power"][0] - expected) / expected < 1e-5
    assert abs(c["deposited_power"][0] - ref["deposited_power"][0]) / expected < 1e-9
    # two half ranges accumulate to the same totals
    tracer.reset()
    tracer.trace(sink_mode=1, first_photon=0, photon_count=n // 2)
    tracer.trace(sink_mode=1, first_photon=n // 2, photon_count=n - n // 2)
    c2 = tracer.counters()
    assert c2["bounces"] == c["bounces"] and c2["photons"] == n
    assert abs(c2["deposited_power"][0] - c["deposited_power"][0]) / expected < 1e-9


@pytest.mark.parametrize("photons", [1, 63, 65, 4097])
def test_ragged_passes_equal_oracle(rt, scene_opener, photons):
    """Passes that do not fill a wave, a run or the grid (three lights, so the id space is ragged
    as well): paths and counters as the oracle's."""
    entry = dict(SCENES["mixed"], photons_per_light=photons)
    scene, tracer, orc = _open(rt, scene_opener, entry)
    total = photons * scene.desc.light_count
    tracer.reset()
    tracer.trace(sink_mode=0, record_photons=total)
    got, c = tracer.records(), tracer.counters()
    ref = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=2, record_photons=total)
    assert c["photons"] == total == ref["photons"]
    for k in ("casts", "bounces", "deposits"):
        assert c[k] == ref[k], k
    assert_bits_equal(got["emitted"], ref["emitted"], "emitted")
    assert_bits_equal(got["bounce_f64"], ref["bounce_f64"], "hits")
    assert np.array_equal(got["bounce_i32"], ref["bounce_i32"]), "ids"


@pytest.mark.parametrize("name", ["bombardment", "mixed"])
def test_high_photon_indices_equal_oracle(rt, scene_opener, name):
    """The last rank of an eight-GPU weak-scaling frame traces photon indices around 1.4e9 of a
    1.6e9-photon budget (beyond 2^30, and for three lights ids beyond 2^32): paths and counters
    of such a range are the oracle's."""
    entry = dict(SCENES[name], photons_per_light=1_600_000_000)
    scene, tracer, orc = _open(rt, scene_opener, entry)
    first, count = 1_400_000_017, 3000
    total = count * scene.desc.light_count
    tracer.reset()
    tracer.trace(sink_mode=0, first_photon=first, photon_count=count, record_photons=total)
    got, c = tracer.records(), tracer.counters()
    ref = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=2, first_photon=first, photon_count=count,
                    record_photons=total)
    assert c["photons"] == total == ref["photons"]
    for k in ("casts", "bounces", "deposits"):
        assert c[k] == ref[k], k
    assert ref["bounces"] > 0
    assert_bits_equal(got["emitted"], ref["emitted"], "emitted")
    assert_bits_equal(got["bounce_f64"], ref["bounce_f64"], "hits")
    assert np.array_equal(got["bounce_i32"], ref["bounce_i32"]), "ids"
    assert np.array_equal(got["bounce_power"], ref["bounce_power"]), "power"


def test_collective_entry_point_with_one_rank(rt, scene_opener, tracer_mod):
    """rt_allreduce over a communicator of this one GPU: RCCL loads, the all-reduce runs on the
    tracer's stream over the camera buffers and the counters, and with one rank changes nothing."""
    scene, tracer, orc = _open(rt, scene_opener, dict(SCENES["multicam"], photons_per_light=20000))
    tracer.set_splat(0.05)
    tracer.reset()
    tracer.trace(sink_mode=2)
    before = [tracer.camera_buffer(c).copy() for c in range(scene.desc.camera_count)]
    mine = tracer.counters()
    assert mine["splat_total"] > 0 and any(b.any() for b in before)
    (comm,) = tracer_mod.comm_init_all([0])
    try:
        totals = tracer.allreduce(comm)
    finally:
        tracer_mod.comm_destroy(comm)
    for name, value in mine.items():
        assert np.array_equal(totals[name], value), name
    for c, b in enumerate(before):
        assert_bits_equal(tracer.camera_buffer(c), b, f"camera {c} after a one-rank all-reduce")


def test_every_rank_issues_the_same_collectives(tmp_path):
    """Ranks of a communicator must issue the same collectives in the same order or they deadlock.
    What rt_allreduce issues is read through a logging stand-in for librccl.so (tests/stubs) in a
    process of its own: the sequence must not depend on whether the caller asked for the totals, and
    must be one all-reduce per camera buffer, then the integer counters, then the three powers."""
    import subprocess
    import sys
    stubs = os.path.join(REPO, "tests", "stubs")
    assert os.path.exists(os.path.join(stubs, "librccl.so")), "build the stand-in with __graft_entry__.build()"
    script = """
import importlib, os, sys
sys.path.insert(0, {repo!r})
rt = importlib.import_module("ray-tracer_amd")
tracer_mod = importlib.import_module("ray-tracer_amd.tracer")
scene = rt.Scene(os.path.join({repo!r}, "scenes", "MultiCameraTest.json"))
scene.settings.photons_per_light = 5000
tracer = rt.Tracer(scene)
tracer.set_splat(0.05)
tracer.reset()
tracer.trace(sink_mode=2)
(comm,) = tracer_mod.comm_init_all([0])
mine = tracer.counters()
for want_totals in (True, False):
    os.environ["RCCL_STUB_LOG"] = sys.argv[1] + (".totals" if want_totals else ".null")
    totals = tracer.allreduce(comm, want_totals=want_totals)
    if want_totals:
        assert all((totals[k] == v) if k != "deposited_power" else (totals[k] == v).all() for k, v in mine.items())
print(scene.desc.camera_count, " ".join(str(scene.desc.cameras[c].width * scene.desc.cameras[c].height * 3)
                                       for c in range(scene.desc.camera_count)))
""".format(repo=REPO)
    env = dict(os.environ, LD_LIBRARY_PATH=stubs + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    log = str(tmp_path / "rccl")
    r = subprocess.run([sys.executable, "-c", script, log], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    cameras, *floats = r.stdout.split()
    with_totals = open(log + ".totals").read().splitlines()
    without = open(log + ".null").read().splitlines()
    assert with_totals == without
    expected = [f"allreduce {n} 7 0" for n in floats] + ["allreduce 15 5 0", "allreduce 3 8 0"]
    assert int(cameras) == len(floats) >= 2 and with_totals == expected


def test_empty_inputs(rt, scene_opener):
    """Nothing to trace, nothing to hit, nothing to keep: no crash, no hang, the obvious counters."""
    entry = dict(SCENES["mixed"], photons_per_light=2000)
    scene, tracer, orc = _open(rt, scene_opener, entry)
    lights = scene.desc.light_count
    # an empty photon range
    tracer.reset()
    tracer.trace(sink_mode=0, first_photon=2000, photon_count=0)
    c = tracer.counters()
    assert c["photons"] == 0 and c["casts"] == 0
    # no probes: every deposit is culled, the store stays empty
    tracer.set_probes(np.zeros((0, 3)), 10.0, 10.0, 64)
    tracer.reset()
    tracer.trace(sink_mode=1)
    c = tracer.counters()
    assert c["photons"] == 2000 * lights and c["kept"] == 0 and c["store_size"] == 0 and c["culled"] == c["deposits"] > 0
    assert tracer.store().shape[0] == 0
    # a scene without volumes: every photon escapes on its first cast
    scene.desc.volume_count = 0
    bare = rt.Tracer(scene)
    bare.reset()
    bare.trace(sink_mode=0)
    c = bare.counters()
    assert c["photons"] == c["casts"] == 2000 * lights and c["bounces"] == 0
    rays = np.array([[0.0, 0.0, 0.0, 0.0, 0.0, 1.0]])
    _, ids = bare.cast_rays(rays)
    assert ids[0, 0] == 0


def test_errors_are_codes_not_crashes(rt, scene_opener, tracer_mod):
    scene, tracer, _ = _open(rt, scene_opener, SCENES["furnace"])
    with pytest.raises(tracer_mod.TracerError) as e:
        tracer.trace(sink_mode=1)  # probes not set
    assert e.value.code == 3
    with pytest.raises(tracer_mod.TracerError) as e:
        tracer.camera_buffer  # attribute exists
        tracer_mod._check(tracer_mod.tracer_lib().rt_read_camera_buffer(tracer._handle, 7, None))
    assert e.value.code == 1
    with pytest.raises(tracer_mod.TracerError):
        rt.Tracer(scene, device=99)


def test_store_limit_is_the_store_capacity_of_a_rank(rt, scene_opener):
    """rt_set_store_limit: the store takes that many records in all and counts the rest as dropped, as the
    reference's store does at its capacity (src/BounceStore.cpp:21-33); 0 gives the capacity back; a limit
    below what the store holds is refused; the records kept under a limit are records of the full run."""
    entry = {"scene": "tests/scenes/mixed.json", "photons_per_light": 3000}
    scene, tracer, orc = _open(rt, scene_opener, entry)
    setup = O.frame_setup(orc, O.RNG_COUNTER, O.MATH_PORTABLE)
    keep = setup["keep_radius"]
    tracer.set_probes(tracer.collect_probe_positions(), keep, keep, setup["store_capacity"])
    tracer.reset()
    tracer.trace(sink_mode=1)
    full = tracer.counters()
    everything = _sorted_records(tracer.store())
    assert full["store_dropped"] == 0 and full["store_size"] > 40
    limit = full["store_size"] // 3
    with pytest.raises(RuntimeError):
        tracer.set_store_limit(limit)  # the store holds more than that
    tracer.reset()
    tracer.set_store_limit(limit)
    tracer.trace(sink_mode=1)
    part = tracer.counters()
    assert part["store_size"] == limit
    assert part["store_attempted"] == full["store_attempted"]
    assert part["store_dropped"] == full["store_attempted"] - limit
    assert part["bounces"] == full["bounces"] and part["kept"] == full["kept"]
    kept = tracer.store()
    assert kept.shape[0] == limit
    rows = {r.tobytes() for r in np.ascontiguousarray(everything)}
    assert all(np.ascontiguousarray(r).tobytes() in rows for r in kept)
    tracer.reset()
    tracer.set_store_limit(0)
    tracer.trace(sink_mode=1)
    again = tracer.counters()
    assert again["store_size"] == full["store_size"] and again["store_dropped"] == 0
