"""Frame-level glue around the tracer: sink set-up and the sharded frame as Renderer::renderFrame
does them (reference src/Renderer.cpp:222-419)."""
import ctypes as C


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


def share_store(tracer, capacity, emitter_spacing, rank, world):
    """Limit this rank's store to its share of the frame's one store (parallel.store_share).  Every
    rank lays down the emitter deposits once to learn how many there are (they depend on the probes
    only, which the ranks share); they stay in rank 0's store and leave the others'.  Call after
    set_probes* and reset; returns the number of emitter deposits."""
    from . import parallel

    _, emitters = tracer.deposit_emitters(emitter_spacing)
    if rank != 0:
        tracer.reset()
    tracer.set_store_limit(parallel.store_share(capacity, emitters, rank, world))
    return emitters


def render_frame(rt, scene, tracer, buffers, rank=0, world=1, dist=None):
    """One frame on this rank's GPU, reduced over the ranks: Renderer::renderFrame (reference
    src/Renderer.cpp:222-700) with the photon indices sharded.

    `buffers[c]` is the torch tensor bound as camera c's buffer (bind_camera_buffer), so that the
    collective reduces it in place.  Probe-gather mode: every rank runs the probe pass and builds
    the same keep-test index, rank 0 alone keeps the emitter faces' deposits (they must enter the sum
    once), every rank traces its share into its own store — limited to its part of the frame's one
    store, so that an overflowing frame keeps as many records as on one GPU — and gathers it into its own camera
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
        _, capacity = tracer.set_probes_from_cameras(d["keep_radius"], d["keep_radius"], 256,
                                                     int(scene.settings.bounce_store_capacity))
        tracer.reset()
        if world > 1:
            share_store(tracer, capacity, max(cell * 0.5, 1e-6), rank, world)  # (leaves rank 0's emitter deposits in)
        elif rank == 0:
            tracer.deposit_emitters(max(cell * 0.5, 1e-6))
        if count > 0:  # an empty share traces nothing (a count of 0 would mean "all the rest")
            tracer.trace(sink_mode=rt.tracer.SINK_PROBE, first_photon=first, photon_count=count)
        for c in gathering:
            tracer.gather(c, cell, d["min_splat_radius"])
    else:
        prepare_sinks(rt, scene, tracer, "splat")
        tracer.reset()
        if count > 0:
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
