"""Frame-level glue around the tracer: sink set-up as Renderer::renderFrame does it
(reference src/Renderer.cpp:222-419) and the CPU baseline leg of bench.py."""
import ctypes as C
import json
import os
import subprocess
import tempfile
import time

import numpy as np


def derive(rt, scene):
    """sceneDepthFootprint, cellSize, minSplatRadius, keepRadius (reference
    src/Renderer.cpp:222-274,303-310), computed by the C++ host library."""
    out = (C.c_double * 4)()
    lib = rt.host_lib()
    lib.rth_frame_derive.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    if lib.rth_frame_derive(C.cast(scene.ptr, C.c_void_p), out) != 0:
        raise RuntimeError(lib.rth_last_error().decode())
    return dict(scene_depth_footprint=out[0], cell_size=out[1], min_splat_radius=out[2], keep_radius=out[3])


def prepare_sinks(rt, scene, tracer, sink):
    """Configure the tracer's sink for a frame; returns what was derived."""
    d = derive(rt, scene)
    info = dict(d)
    if sink == "splat":
        # the reference's DensityGrid grows without bound; size ours for one cell per deposit, 4 Mi .. 128 Mi cells
        bound = int(scene.settings.photons_per_light) * max(1, int(scene.desc.light_count)) * (
            max(0, int(scene.settings.bounce_threshold)) + 1)
        tracer.set_splat(d["min_splat_radius"], d["cell_size"], min(1 << 27, max(1 << 22, bound)))
    elif sink == "probe":
        probes = tracer.collect_probe_positions()
        # store sizing, reference src/Renderer.cpp:382-390
        capacity = int(scene.settings.bounce_store_capacity)
        if probes.shape[0]:
            capacity = min(capacity, max(1, probes.shape[0] * 256))
        tracer.set_probes(probes, d["keep_radius"], d["keep_radius"], capacity)
        info.update(probe_count=int(probes.shape[0]), store_capacity=capacity, probes=probes)
    return info


def render_frame(rt, scene, tracer, buffers, rank=0, world=1, dist=None):
    """One frame on this rank's GPU, reduced over the ranks: Renderer::renderFrame (reference
    src/Renderer.cpp:222-700) with the photon indices sharded.

    `buffers[c]` is the torch tensor bound as camera c's buffer (bind_camera_buffer), so that the
    collective reduces it in place.  Probe-gather mode: every rank runs the probe pass and builds
    the same keep-test index, rank 0 alone deposits the emitter faces (they must enter the sum
    once), every rank traces its share into its own store and gathers it into its own camera
    buffers — the gather is linear in the deposits (reference src/ProbeGather.cpp:653-727) — and
    one all-reduce(sum) per camera makes the frame.  Splat mode: the trace accumulates into the
    buffers directly, then the same all-reduce.  Returns the reduced counters."""
    from . import parallel

    d = derive(rt, scene)
    photons = int(scene.settings.photons_per_light)
    first, count = parallel.shard_range(photons, rank, world)
    cameras = range(scene.desc.camera_count)
    gathering = [c for c in cameras
                 if scene.desc.cameras[c].bounce_filter < 0 and scene.desc.cameras[c].light_filter < 0]
    if scene.settings.use_probe_gather:
        cell = d["scene_depth_footprint"] if d["scene_depth_footprint"] > 0.0 else d["cell_size"]
        tracer.set_probes_from_cameras(d["keep_radius"], d["keep_radius"], 256, int(scene.settings.bounce_store_capacity))
        tracer.reset()
        if rank == 0:
            tracer.deposit_emitters(max(cell * 0.5, 1e-6))
        tracer.trace(sink_mode=rt.tracer.SINK_PROBE, first_photon=first, photon_count=count)
        for c in gathering:
            tracer.gather(c, cell, d["min_splat_radius"])
    else:
        prepare_sinks(rt, scene, tracer, "splat")
        tracer.reset()
        tracer.trace(sink_mode=rt.tracer.SINK_SPLAT, first_photon=first, photon_count=count)
    counters = tracer.counters()  # waits for the stream
    parallel.all_reduce_sum(list(buffers), dist)
    names = ("photons", "casts", "bounces", "deposits", "kept", "culled", "store_attempted", "store_size", "splat_total")
    if world > 1:
        import torch
        totals = torch.tensor([int(counters[k]) for k in names], dtype=torch.int64, device=buffers[0].device)
        parallel.all_reduce_sum([totals], dist)
        counters = dict(counters, **dict(zip(names, (int(v) for v in totals.tolist()))))
    return counters


def cpu_baseline(rt, scene_path, sink, setup, photons):
    """Time the CPU path on a bounded sample of the same workload (same scene, `photons`
    photons per light instead of the full budget; cost is linear in photons).

    Preferred: the REFERENCE itself (oracle/_ref/refdump bench, its own multithreaded
    Worker loop, photonPassSeconds as it reports it); its bounce count comes from the
    oracle run at the same photon count (the reference does not count delta bounces).
    Fallback: the oracle port."""
    import oracle_lib as O

    cores = min(16, os.cpu_count() or 1)
    scene = rt.Scene(scene_path)
    scene.settings.photons_per_light = photons
    orc = O.Oracle(scene)
    kwargs = {}
    if sink == "probe":
        kwargs = dict(sink_mode=O.SINK_PROBE, probes=setup["probes"], probe_cell_size=setup["keep_radius"],
                      keep_radius=setup["keep_radius"], store_capacity=setup["store_capacity"])
    elif sink == "splat":
        kwargs = dict(sink_mode=O.SINK_SPLAT, min_splat_radius=setup["min_splat_radius"],
                      density_cell_size=setup["cell_size"])
    t0 = time.perf_counter()
    r = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, threads=cores, **kwargs)
    port_seconds = time.perf_counter() - t0
    result = {
        "value": r["bounces"] / port_seconds, "unit": "photon-bounces/s", "cores": cores, "kind": "port",
        "sample": f"{photons} photons per light of the same scene ({r['bounces']} bounces, {port_seconds:.2f} s), "
                  f"oracle/liboracle.so with {cores} threads",
    }
    refdump = os.path.join(rt.REPO_ROOT, "oracle", "_ref", "refdump")
    if os.path.exists(refdump):
        try:
            with tempfile.TemporaryDirectory() as tmp:
                blob = os.path.join(tmp, "scene.rtsb")
                scene.settings.use_probe_gather = 1 if sink == "probe" else 0
                scene.settings.worker_count = cores
                scene.write(blob)
                proc = subprocess.run([refdump, "bench", blob, str(cores)], capture_output=True, text=True,
                                      timeout=600, check=True)
                line = [l for l in proc.stdout.splitlines() if l.startswith("{")][-1]
                ref = json.loads(line)
                result = {
                    "value": r["bounces"] / ref["photon_pass_seconds"], "unit": "photon-bounces/s", "cores": cores,
                    "kind": "reference",
                    "sample": f"{photons} photons per light of the same scene; the reference's own Worker path "
                              f"(oracle/_ref/refdump bench, {cores} workers) took photonPassSeconds="
                              f"{ref['photon_pass_seconds']:.2f} (includes its single-threaded probe pass); "
                              f"{r['bounces']} bounces counted by the oracle at the same photon count",
                    "port_value": r["bounces"] / port_seconds,
                }
        except Exception as e:  # the reference binary is optional on the GPU box
            result["reference_error"] = str(e)[:200]
    return result
