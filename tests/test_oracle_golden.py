"""Pin the CPU oracle to the reference.

Every expected value here was produced by the reference's own compiled code
(oracle/_ref/refdump via oracle/make_golden.py); the oracle must reproduce it bit
for bit when it uses the reference's arithmetic (mt19937 stream + libm).
"""
import numpy as np
import pytest

import oracle_lib as O
from conftest import assert_bits_equal

UNIT = [
    ("tri", "orc_intersect_triangles", 8, np.float64),
    ("bounds", "orc_intersect_bounds", 0, np.int32),
    ("sphere", "orc_intersect_spheres", 8, np.float64),
    ("quad", "orc_intersect_quads", 20, np.float64),
    ("plane", "orc_intersect_planes", 8, np.float64),
    ("quat", "orc_quaternions", 10, np.float64),
    ("eval", "orc_bsdf_evaluate", 4, np.float64),
]


@pytest.mark.parametrize("name,fn,cols,dtype", UNIT)
def test_function_level_vectors(golden, name, fn, cols, dtype):
    g = golden("unit")
    got = O.batch(fn, g[f"{name}_in"], cols, dtype)
    assert_bits_equal(got, g[f"{name}_out"].astype(dtype), name)


def test_watertight_diagonal_has_exactly_one_owner(golden):
    """tests/test_TriangleWatertight.cpp:52-75,97-128 — re-stated on the oracle's output."""
    g = golden("unit")
    hit = O.batch("orc_intersect_triangles", g["tri_in"][:804], 8)[:, 0].reshape(201, 4)
    interior = hit[1:200]
    assert (interior[:, 0] + interior[:, 1] == 1).all()  # axis-aligned sweep
    assert (interior[:, 2] + interior[:, 3] == 1).all()  # oblique sweep
    back = O.batch("orc_intersect_triangles", g["tri_in"][804:806], 8)[:, 0]
    assert (back == 0).all()  # back-face cull (:130-139)


def test_bsdf_sample_vectors(golden):
    g = golden("unit")
    got = O.batch("orc_bsdf_sample", g["bsdf_in"], 10, extra=(O.C.c_int32(O.MATH_LIBM),))
    assert_bits_equal(got, g["bsdf_out"], "bsdf sample")
    # the portable sin/cos moves sampled directions by at most a few ulp
    portable = O.batch("orc_bsdf_sample", g["bsdf_in"], 10, extra=(O.C.c_int32(O.MATH_PORTABLE),))
    assert np.array_equal(portable[:, 0], got[:, 0])
    assert np.abs(portable[:, 1:4] - got[:, 1:4]).max() < 1e-15


def test_bsdf_properties(golden):
    """tests/test_BSDF.cpp: Lambert weight == albedo (:101-127), VNDF weight <= 1 (:215-266),
    mirror direction (:53-86)."""
    g = golden("unit")
    rows, out = g["bsdf_in"], g["bsdf_out"]
    valid = out[:, 0] == 1
    lambert = (rows[:, 0] == 0) & valid
    assert np.array_equal(out[lambert, 4:7], rows[lambert, 1:4].astype(np.float32).astype(np.float64))
    micro = (rows[:, 0] == 3) & valid
    assert micro.sum() > 100 and out[micro, 4:7].max() <= 1.0 + 1e-6
    mirror = (rows[:, 0] == 1) & valid
    d, n, i = out[mirror, 1:4], rows[mirror, 9:12], rows[mirror, 6:9]
    expect = i - 2 * (i * n).sum(1, keepdims=True) * n
    assert np.abs(d - expect / np.linalg.norm(expect, axis=1, keepdims=True)).max() < 1e-9
    assert np.abs(np.linalg.norm(out[valid, 1:4], axis=1) - 1).max() < 1e-9


def test_emission_vectors(rt, golden, manifest, scene_opener):
    entry = manifest["emit"]
    g = golden("emit")
    scene = scene_opener(entry)
    orc = O.Oracle(scene)
    n = entry["n"]
    for light in range(scene.desc.light_count):
        rays, colors = orc.emit_mt(light, n, entry["seed"], n)
        assert_bits_equal(rays, g[f"emit_L{light}_rays"], f"light {light} rays")
        assert_bits_equal(colors, g[f"emit_L{light}_color"], f"light {light} color")


@pytest.mark.parametrize("key", ["walk_furnace", "walk_mixed", "walk_animated", "walk_singlephoton", "walk_mirrordirect",
                                 "walk_cbgmd", "walk_stochastic128"])
def test_seeded_walk_matches_reference(golden, manifest, scene_opener, key):
    """One seeded worker: every bounce's hit (position, normal, distance bits), volume,
    triangle id, material, photon power and every emitted ray equal the reference's."""
    entry = manifest[key]
    g = golden(key)
    scene = scene_opener(entry)
    orc = O.Oracle(scene)
    derived = orc.derive()
    r = orc.trace(rng_mode=O.RNG_MT, math_mode=O.MATH_LIBM, sink_mode=O.SINK_SPLAT,
                  record_photons=entry["record"], min_splat_radius=derived["min_splat_radius"])
    totals = g["walk_totals"]
    assert [r["photons"], r["casts"], r["bounces"], r["deposits"]] == [int(v) for v in totals[:4]]
    assert_bits_equal(r["emitted"], g["walk_emitted"], "emitted rays")
    assert np.array_equal(r["casts_valid"], g["walk_casts"])
    f64, i32, power = O.compact_bounces(r)
    assert_bits_equal(f64, g["walk_bounce_f64"], "bounce hits")
    assert_bits_equal(power, g["walk_bounce_power"], "bounce power")
    ref_i32 = g["walk_bounce_i32"]
    cols = [0, 1, 2, 4, 6, 7]  # photon, bounce, volume, material, delta, light
    assert np.array_equal(i32[:, cols], ref_i32[:, cols])
    unambiguous = ref_i32[:, 5] <= 1
    assert np.array_equal(i32[unambiguous, 3], ref_i32[unambiguous, 3]), "triangle ids"
    # derived frame quantities (Renderer.cpp:222-274,303-310)
    assert_bits_equal(np.array([derived[k] for k in ("scene_depth_footprint", "cell_size", "min_splat_radius",
                                                      "keep_radius")]), totals[10:14], "derived")
    # Worker::splatToCamera: buffers bitwise, counters equal
    for c, buf in enumerate(r["splat"]):
        assert_bits_equal(buf, g[f"walk_splat_c{c}"], f"splat buffer {c}")
    assert [r["splat_total"], r["splat_radius_clamped"], r["splat_luminance_clamped"]] == [int(v) for v in totals[7:10]]
    assert np.allclose(r["deposited_power"], totals[4:7], rtol=1e-12)


def test_render_frame_store_matches_reference(golden, manifest, scene_opener):
    """Renderer::renderFrame's BounceStore (default probe mode): same records, same order."""
    entry = manifest["render_furnace"]
    g = golden("render_furnace")
    assert g["render_crosscheck"][1] == 0  # the dump's own walk-vs-Worker check
    scene = scene_opener(entry)
    orc = O.Oracle(scene)
    derived = orc.derive()
    # the furnace keeps every bounce (keep radius 1e5); one probe anywhere inside suffices
    r = orc.trace(rng_mode=O.RNG_MT, math_mode=O.MATH_LIBM, sink_mode=O.SINK_PROBE,
                  probes=np.zeros((1, 3)), probe_cell_size=derived["keep_radius"],
                  keep_radius=derived["keep_radius"], store_capacity=int(g["render_counters"][5]))
    counters = g["render_counters"]
    assert [r["kept"], r["culled"], r["store_dropped"], r["store_attempted"], r["store_size"]] == \
        [int(v) for v in counters[:5]]
    head = g["render_store_head"]
    assert_bits_equal(r["store"][:head.shape[0]], head, "store records")
    assert np.allclose(r["store"][:, 10].astype(np.float64).sum(), g["render_ledger"][1], rtol=1e-12)


def test_furnace_ledger_closed_form(manifest, scene_opener):
    """tests/test_FurnaceLedger.cpp:35-70: sum of deposited power == Phi (1 - rho^(B+1)) / (1 - rho),
    store size == N (B+1), nothing culled or dropped."""
    scene = scene_opener({"scene": "tests/scenes/furnace.json", "photons_per_light": 20000})
    orc = O.Oracle(scene)
    s = scene.settings
    r = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, sink_mode=O.SINK_PROBE, threads=4,
                  probes=np.zeros((1, 3)), probe_cell_size=1e5, keep_radius=1e5, store_capacity=1 << 20)
    phi, rho, terms = 100.0 * 4 * np.pi, 0.5, s.bounce_threshold + 1
    expected = phi * (1 - rho ** terms) / (1 - rho)
    assert r["store_size"] == 20000 * terms and r["culled"] == 0 and r["store_dropped"] == 0
    assert abs(r["store"][:, 10].astype(np.float64).sum() - expected) / expected < 1e-4
    assert abs(r["deposited_power"][0] - expected) / expected < 1e-4


def test_counter_mode_is_thread_count_invariant(manifest, scene_opener):
    """Per-photon Philox streams: 1 thread and 4 threads trace identical paths."""
    scene = scene_opener(manifest["walk_mixed"])
    orc = O.Oracle(scene)
    a = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=1, record_photons=2000)
    b = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=4, record_photons=2000)
    for k in ("emitted", "bounce_f64", "bounce_power"):
        assert_bits_equal(a[k], b[k], k)
    assert np.array_equal(a["bounce_i32"], b["bounce_i32"])
    assert [a[k] for k in ("photons", "casts", "bounces", "deposits")] == \
        [b[k] for k in ("photons", "casts", "bounces", "deposits")]
    # sub-ranges tile the full range (the multi-GPU partition)
    lo = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, first_photon=0, photon_count=150)
    hi = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, first_photon=150, photon_count=250)
    assert lo["bounces"] + hi["bounces"] == a["bounces"] and lo["casts"] + hi["casts"] == a["casts"]


def test_decay_and_cap_termination(scene_opener):
    """tests/test_DecayTermination.cpp, tests/test_SelfHitLive.cpp:37-63: with floor 0 only the cap
    terminates -> exactly B+1 deposits per photon in the sealed box; a floor above the emitted
    magnitude stops every photon after its first deposit."""
    scene = scene_opener({"scene": "tests/scenes/furnace.json", "photons_per_light": 2000, "bounce_threshold": 3})
    orc = O.Oracle(scene)
    r = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, record_photons=2000)
    assert (r["casts_valid"] == 4).all()
    scene.settings.termination_threshold = 1e9
    orc2 = O.Oracle(scene)
    r2 = orc2.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, record_photons=2000)
    assert (r2["casts_valid"] == 1).all()


@pytest.mark.parametrize("key", ["probes_mixed", "probes_bombardment", "probes_animated", "probes_lens"])
def test_probe_pass_matches_reference(golden, manifest, scene_opener, key):
    """ProbeGather::collectGatherPoints: every gather-point record (pixel, position, normal,
    view direction, material, throughput, path length, footprint, time, weight) and the three
    counters equal the reference's bit for bit (mt19937 draws, libm), camera by camera."""
    g = golden(key)
    scene = scene_opener(manifest[key])
    orc = O.Oracle(scene)
    for c in range(scene.desc.camera_count):
        rows, stats = orc.collect_gather_points(c, rng_mode=O.RNG_MT, math_mode=O.MATH_LIBM)
        want = g[f"probes_c{c}"]
        assert want.shape[0] > 0
        assert_bits_equal(rows, want, f"{key} camera {c}")
        counts = g[f"probes_counts_c{c}"]
        assert [stats["camera_rays"], stats["delta_extensions"], stats["misses"]] == [int(v) for v in counts[:3]]


def test_probe_pass_counter_mode_differs_only_in_stochastic_samples(manifest, scene_opener):
    """With the per-pixel Philox streams only the samples that draw random numbers
    (dielectric extensions) may differ from the sequential mt19937 pass."""
    scene = scene_opener(manifest["probes_mixed"])
    orc = O.Oracle(scene)
    a, sa = orc.collect_gather_points(0, rng_mode=O.RNG_MT, math_mode=O.MATH_LIBM)
    b, sb = orc.collect_gather_points(0, rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE)
    assert sa["camera_rays"] == sb["camera_rays"]
    # pixels whose samples are all first-hit records (weight 1) are RNG-free
    direct_a, direct_b = a[a[:, 18] == 1.0], b[b[:, 18] == 1.0]
    common = min(len(direct_a), len(direct_b))
    assert common > 0
    key_a = {(int(r[0]), int(r[1])): r for r in direct_a}
    same = 0
    for r in direct_b:
        q = key_a.get((int(r[0]), int(r[1])))
        if q is not None and q[11] == r[11] and np.allclose(q[2:5], r[2:5], rtol=0, atol=1e-9):
            same += 1
    assert same > 0.9 * common


@pytest.mark.parametrize("key", ["render_mixed", "render_mixed_tight", "render_animated", "render_lens"])
def test_whole_probe_mode_frame_matches_reference(golden, manifest, scene_opener, key):
    """Renderer::renderFrame in deterministic probe mode, stage by stage against the reference's
    own output: emitter deposits + photon deposits (the store, record for record), every
    camera's gathered float buffer and its tonemapped 16-bit image, bit for bit."""
    g = golden(key)
    scene = scene_opener(manifest[key])
    orc = O.Oracle(scene)
    setup = O.frame_setup(orc, O.RNG_MT, O.MATH_LIBM)
    counters = g["render_counters"]
    assert setup["store_capacity"] == int(counters[5])
    emitters, estats = orc.deposit_emitters(setup["probes"], setup["keep_radius"], setup["keep_radius"],
                                            setup["emitter_spacing"])
    assert estats["kept"] == int(counters[6]) and estats["kept"] > 0
    r = orc.trace(rng_mode=O.RNG_MT, math_mode=O.MATH_LIBM, sink_mode=O.SINK_PROBE, probes=setup["probes"],
                  probe_cell_size=setup["keep_radius"], keep_radius=setup["keep_radius"],
                  store_capacity=setup["store_capacity"], preload=emitters)
    assert (r["kept"], r["culled"]) == (int(counters[0]), int(counters[1]))
    assert_bits_equal(r["store"], g["render_store"], "store")
    desc = scene.desc
    for c in range(desc.camera_count):
        cam = desc.cameras[c]
        want = g[f"render_buffer_c{c}"]
        if cam.bounce_filter >= 0 or cam.light_filter >= 0:
            got = np.zeros_like(want)
        else:
            got, stats = orc.gather(c, r["store"], setup["gather_cell"], setup["min_gather_radius"], O.RNG_MT,
                                    O.MATH_LIBM)
            hit, gathered, accum = (int(v) for v in g[f"render_gather_c{c}"][:3])
            assert (stats["pixels_hit"], stats["pixels_gathered"], stats["deposits_accum"]) == (hit, gathered, accum)
            assert gathered > 0
        assert_bits_equal(got, want, f"{key} camera {c} buffer")
        image = O.tonemap(got, float(g[f"render_gather_c{c}"][3]))
        assert np.array_equal(image, g[f"render_image_c{c}"]), "tonemapped image"


def test_first_records_in_photon_order_make_a_reference_frame(golden, manifest, scene_opener):
    """A store that overflows holds, in the reference, the deposits of the photons its workers met first
    (reference src/BounceStore.cpp:11-33, src/Worker.cpp:429-505).  With ten workers on batches of 100 000
    that is "the lowest photon indices" up to a ragged edge.  The oracle's single thread keeps exactly the
    first records in photon order; its frame of the overflowing headline scene (store 4.3 times over, as at
    full size) must be statistically one more of the reference's frames — which is what entitles the device
    to settle an overflow by photon order (tests/test_gpu_parity.py compares its store with this one bit
    for bit, and its frame with the same fixture)."""
    from conftest import assert_one_more_frame
    key = "render_bombardment_overflow"
    g, entry = golden(key), dict(manifest[key])
    references = [g["render_buffer_c0"].astype(np.float64)] + [g[f"spread{i}_buffer_c0"].astype(np.float64) for i in (1, 2, 3)]
    for name in ("render_counters", "spread1_counters", "spread2_counters", "spread3_counters"):
        assert g[name][4] == g[name][5] == entry["bounce_store_capacity"]    # store size == capacity: full ...
        assert 4.2 < g[name][3] / g[name][5] < 4.35                           # ... attempted / capacity
    scene = scene_opener(entry)
    orc = O.Oracle(scene)
    setup = O.frame_setup(orc, O.RNG_COUNTER, O.MATH_PORTABLE)
    assert setup["store_capacity"] == entry["bounce_store_capacity"]
    keep = setup["keep_radius"]
    # the store fills up within the first quarter of the photons: no later photon can change it
    r = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, sink_mode=O.SINK_PROBE, threads=1, probes=setup["probes"],
                  probe_cell_size=keep, keep_radius=keep, store_capacity=setup["store_capacity"],
                  photon_count=entry["photons_per_light"] * 3 // 10)
    assert r["store_size"] == setup["store_capacity"] and r["store_dropped"] > 0
    frame, _ = orc.gather(0, r["store"], setup["gather_cell"], setup["min_gather_radius"])
    assert_one_more_frame(frame.astype(np.float64), references)
