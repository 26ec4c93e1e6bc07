"""How unevenly a frame's gather is loaded: records in the 27 cells around each gather point (what the wave
that takes the pixel has to look at) for the first camera of the given scenes, each at its own budget.

    python tools/gather_load.py <scene name> ...
"""
import importlib, os, sys
import numpy as np

sys.path.insert(0, os.getcwd())
rt = importlib.import_module("ray-tracer_amd")
frame = importlib.import_module("ray-tracer_amd.frame")
for name in sys.argv[1:]:
    scene = rt.Scene(f"scenes/{name}.json")
    tracer = rt.Tracer(scene, device=0)
    setup = frame.prepare_sinks(rt, scene, tracer, "probe")
    cell = setup["scene_depth_footprint"] if setup["scene_depth_footprint"] > 0.0 else setup["cell_size"]
    tracer.reset()
    tracer.deposit_emitters(max(cell * 0.5, 1e-6))
    tracer.trace(sink_mode=rt.tracer.SINK_PROBE)
    store = tracer.store()
    pts, _ = tracer.collect_gather_points(0)
    rc = np.floor(store[:, 0:3].astype(np.float64) / cell).astype(np.int64)
    pc = np.floor(pts[:, 2:5] / cell).astype(np.int64)
    lo = np.minimum(rc.min(axis=0), pc.min(axis=0)) - 1
    dim = np.maximum(rc.max(axis=0), pc.max(axis=0)) + 2 - lo
    key = lambda c: ((c[:, 2] - lo[2]) * dim[1] + (c[:, 1] - lo[1])) * dim[0] + (c[:, 0] - lo[0])
    cells, counts = np.unique(key(rc), return_counts=True)
    load = np.zeros(len(pc), dtype=np.int64)
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                k = key(pc + np.array([dx, dy, dz]))
                at = np.searchsorted(cells, k)
                at[at >= len(cells)] = 0
                load += np.where(cells[at] == k, counts[at], 0)
    busy = load[load > 0]
    q = np.percentile(busy, [50, 90, 99, 100]).astype(np.int64)
    print(f"{name}: {len(store)} records, {len(pc)} gather points, {len(busy)} with records; records looked at per point: "
          f"mean {busy.mean():.0f}, median {q[0]}, 90% {q[1]}, 99% {q[2]}, most {q[3]}; all points {busy.sum()}; "
          f"the busiest point alone {q[3] / 64:.0f} steps of 64, the mean {busy.mean() / 64:.0f}", flush=True)
