import importlib, sys, time
sys.path.insert(0, '.')
rt = importlib.import_module("ray-tracer_amd")
frame = importlib.import_module("ray-tracer_amd.frame")
scene = rt.Scene("scenes/MirrorKnot4k.json")
tracer = rt.Tracer(scene)
d = frame.derive(rt, scene)
probes, capacity = tracer.set_probes_from_cameras(d["keep_radius"], d["keep_radius"], 256, int(scene.settings.bounce_store_capacity))
tracer.reset()
cell = d["scene_depth_footprint"] if d["scene_depth_footprint"] > 0.0 else d["cell_size"]
print("spacing", max(cell * 0.5, 1e-6), "keep radius", d["keep_radius"], tracer.deposit_emitters(max(cell * 0.5, 1e-6)))
