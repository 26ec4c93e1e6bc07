"""Wall time of every stage of a frame (probe pass + index, reset, emitter deposits, photon pass with its
sort / trace split, gather per camera) for the given scenes, each at its own photon budget.

    python tools/stage_times.py [--splat] [--photons N] scene.json ...
"""
import argparse
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("ray-tracer_amd")
frame = importlib.import_module("ray-tracer_amd.frame")


def timed(tracer, fn):
    tracer.synchronize()
    t0 = time.perf_counter()
    out = fn()
    tracer.synchronize()
    return 1e3 * (time.perf_counter() - t0), out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--splat", action="store_true")
    ap.add_argument("--photons", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("scenes", nargs="+")
    args = ap.parse_args()
    for path in args.scenes:
        scene = rt.Scene(path)
        if args.photons:
            scene.settings.photons_per_light = args.photons
        tracer = rt.Tracer(scene, device=0)
        d = frame.derive(rt, scene)
        name = os.path.basename(path).replace(".json", "")
        cameras = [c for c in range(scene.desc.camera_count)
                   if scene.desc.cameras[c].bounce_filter < 0 and scene.desc.cameras[c].light_filter < 0]
        if args.splat or not scene.settings.use_probe_gather:
            ms_setup, _ = timed(tracer, lambda: frame.prepare_sinks(rt, scene, tracer, "splat"))
            mode = 2
        else:
            ms_setup, _ = timed(tracer, lambda: tracer.set_probes_from_cameras(
                d["keep_radius"], d["keep_radius"], 256, int(scene.settings.bounce_store_capacity)))
            mode = 1
        cell = d["scene_depth_footprint"] if d["scene_depth_footprint"] > 0.0 else d["cell_size"]
        for it in range(args.repeat):
            ms_reset, _ = timed(tracer, tracer.reset)
            ms_emit = 0.0
            if mode == 1:
                ms_emit, _ = timed(tracer, lambda: tracer.deposit_emitters(max(cell * 0.5, 1e-6)))
            ms_trace, _ = timed(tracer, lambda: tracer.trace(sink_mode=mode))
            sort_ms, trace_ms = tracer.timings()
            ms_gather = 0.0
            if mode == 1:
                gstats = []
                for c in cameras:
                    ms, st = timed(tracer, lambda: tracer.gather(c, cell, d["min_splat_radius"]))
                    ms_gather += ms
                    gstats.append(st)
                if it == 0:
                    print(f"{name}: gather cell {cell:.4g} min radius {d['min_splat_radius']:.4g} stats {gstats}", flush=True)
            c = tracer.counters()
            print(f"{name} {'splat' if mode == 2 else 'probe'} it{it}: setup {ms_setup:.2f} reset {ms_reset:.2f} "
                  f"emitters {ms_emit:.2f} pass {ms_trace:.2f} (sort {sort_ms:.2f} trace {trace_ms:.2f}) "
                  f"gather {ms_gather:.2f} ms | bounces {c['bounces']} kept {c['kept']} store {c['store_size']} "
                  f"splats {c['splat_total']} dropped {c['density_dropped']} "
                  f"| pass-only {c['bounces'] / (ms_trace * 1e-3):.3e} bounces/s", flush=True)
        tracer.close()


if __name__ == "__main__":
    main()
