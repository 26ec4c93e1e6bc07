"""rt_trace_static, the entry point that allocates nothing, against rt_trace (MI355X only).

Two calls in a row on one tracer, the second with a smaller input: the static scratch keeps
the first call's contents, so the second one proves that what must start at zero is cleared.
Bar: counters exact, the bounce store equal as a multiset bit for bit, splat buffers to 1e-5 of
the peak (float atomics in a different order).  That the call allocates nothing is checked twice
over: the library counts its own hipMalloc / hipFree calls, and the GPU's free memory
(hipMemGetInfo) is read before and after.
"""
import numpy as np
import pytest

import oracle_lib as O
from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu

COUNTERS = ("photons", "casts", "bounces", "deposits", "kept", "culled", "store_attempted", "store_size",
            "store_dropped")


def _sorted_records(store):
    order = np.lexsort(store.view(np.uint32).T[::-1])
    return store[order]


def _probe_setup(tracer, orc):
    pre = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=8, record_photons=600)
    probes = pre["bounce_f64"][:, 0, :3][pre["casts_valid"][:, 1] > 0][::3]
    tracer.set_probes(probes, 12.0, 12.0, 1 << 20)


@pytest.mark.parametrize("scene_path,photons", [("tests/scenes/mixed.json", (60_000, 7_001)),
                                                ("scenes/CornellBoxGlassMirrorDiffuse.json", (400_000, 90_000))])
def test_static_trace_equals_trace_probe_sink(rt, scene_opener, scene_path, photons):
    scene = scene_opener({"scene": scene_path, "photons_per_light": photons[0] + 11})
    tracer, orc = rt.Tracer(scene), O.Oracle(scene)
    _probe_setup(tracer, orc)
    lights = scene.desc.light_count
    for count in photons:  # the second call is the smaller one
        assert count * lights <= rt.tracer.tracer_lib().rt_trace_static_max_photons()
        tracer.reset()
        tracer.trace(sink_mode=1, first_photon=11, photon_count=count)
        want_counters, want_store = tracer.counters(), _sorted_records(tracer.store())
        tracer.reset()
        tracer.trace_static(sink_mode=1, first_photon=11, photon_count=count)
        got_counters, got_store = tracer.counters(), _sorted_records(tracer.store())
        assert want_counters["photons"] == count * lights and want_counters["kept"] > 0
        for k in COUNTERS:
            assert got_counters[k] == want_counters[k], k
        assert np.allclose(got_counters["deposited_power"], want_counters["deposited_power"], rtol=1e-9)
        assert_bits_equal(got_store, want_store, f"store after {count} photons per light")


def test_static_trace_equals_trace_splat_sink(rt, scene_opener):
    scene = scene_opener({"scene": "scenes/MultiCameraTest.json", "photons_per_light": 300_000})
    tracer, orc = rt.Tracer(scene), O.Oracle(scene)
    d = orc.derive()
    tracer.set_splat(d["min_splat_radius"], d["cell_size"], 1 << 21)
    for count in (300_000, 40_000):
        tracer.reset()
        tracer.trace(sink_mode=2, photon_count=count)
        want, want_buffers = tracer.counters(), [tracer.camera_buffer(c) for c in range(len(tracer.cameras))]
        tracer.reset()
        tracer.trace_static(sink_mode=2, photon_count=count)
        got = tracer.counters()
        for k in ("photons", "casts", "bounces", "deposits", "splat_total", "splat_radius_clamped", "density_cells",
                  "density_deposits", "density_dropped"):
            assert got[k] == want[k], k
        assert want["splat_total"] > 0
        for c, expected in enumerate(want_buffers):
            buffer = tracer.camera_buffer(c)
            assert np.abs(buffer - expected).max() <= 1e-5 * max(expected.max(), 1e-30), f"camera {c}"


def test_static_trace_refuses_what_it_cannot_hold(rt, scene_opener):
    tracer_mod = rt.tracer
    limit = tracer_mod.tracer_lib().rt_trace_static_max_photons()
    scene = scene_opener({"scene": "tests/scenes/furnace.json", "photons_per_light": limit + 1})
    tracer = rt.Tracer(scene)
    tracer.reset()
    with pytest.raises(tracer_mod.TracerError) as e:
        tracer.trace_static(sink_mode=0)
    assert e.value.code == 1 and "exceed" in str(e.value)
    # the largest pass it takes, and the counters of the refused call untouched
    assert tracer.counters()["photons"] == 0
    tracer.trace_static(sink_mode=0, photon_count=limit)
    assert tracer.counters()["photons"] == limit
    with pytest.raises(tracer_mod.TracerError):
        tracer.timings()  # no stage events on this path


def test_static_trace_allocates_nothing(rt, scene_opener):
    """Around rt_trace_static — the very first pass of a fresh tracer, then a smaller one — the library's
    count of device allocations and releases stands still and so does the GPU's free memory."""
    tracer_mod = rt.tracer
    scene = scene_opener({"scene": "tests/scenes/mixed.json", "photons_per_light": 50_000})
    tracer, orc = rt.Tracer(scene), O.Oracle(scene)
    _probe_setup(tracer, orc)  # sink set-up allocates; it comes before
    tracer.reset()
    tracer.synchronize()
    for count in (50_000, 6_000):
        allocations_before = tracer_mod.device_allocations()
        free_before = tracer_mod.device_memory(0)[0]
        tracer.trace_static(sink_mode=1, photon_count=count)
        tracer.synchronize()
        assert tracer_mod.device_allocations() == allocations_before, f"pass of {count}"
        assert tracer_mod.device_memory(0)[0] == free_before, f"pass of {count}"
        counters = tracer.counters()  # reading back must not allocate either
        assert tracer_mod.device_allocations() == allocations_before
    assert counters["photons"] == (50_000 + 6_000) * scene.desc.light_count
    # the counter does count: a pass through rt_trace has to allocate its emission-order lists
    tracer.reset()
    tracer.trace(sink_mode=1, photon_count=6_000)
    assert tracer_mod.device_allocations()[0] > allocations_before[0]
