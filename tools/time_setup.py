"""Wall time and device time of a frame's camera-side set-up (probe pass, probe index, store), twice over:
    python tools/time_setup.py scenes/MirrorKnot4k.json
"""
import importlib
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
rt = importlib.import_module("ray-tracer_amd")
frame = importlib.import_module("ray-tracer_amd.frame")

scene = rt.Scene(sys.argv[1])
t0 = time.perf_counter()
tracer = rt.Tracer(scene)
print(f"rt_create {1e3 * (time.perf_counter() - t0):8.1f} ms")
d = frame.derive(rt, scene)
cams = [c for c in range(scene.desc.camera_count)
        if scene.desc.cameras[c].bounce_filter < 0 and scene.desc.cameras[c].light_filter < 0]
for attempt in range(3):
    tracer.forget_gather_points()
    tracer.synchronize()
    t0 = time.perf_counter()
    for c in cams:
        tracer.count_gather_points(c)
    tracer.synchronize()
    t1 = time.perf_counter()
    probes, capacity = tracer.set_probes_from_cameras(d["keep_radius"], d["keep_radius"], 256, int(scene.settings.bounce_store_capacity))
    tracer.synchronize()
    t2 = time.perf_counter()
    tracer.reset()
    cell = d["scene_depth_footprint"] if d["scene_depth_footprint"] > 0.0 else d["cell_size"]
    tracer.deposit_emitters(max(cell * 0.5, 1e-6))
    tracer.synchronize()
    t3 = time.perf_counter()
    print(f"attempt {attempt}: probe pass {1e3 * (t1 - t0):7.1f} ms  index + store {1e3 * (t2 - t1):7.1f} ms  emitter deposits "
          f"{1e3 * (t3 - t2):6.1f} ms  ({probes} probes, store {capacity}); device: {tracer.setup_timings()}")
