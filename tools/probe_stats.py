"""Probes per keep-test cell of the given scenes (run on the GPU box from the repository root):
    python tools/probe_stats.py MirrorKnot4k CornellBox"""
import importlib, sys, numpy as np
sys.path.insert(0, '.')
rt = importlib.import_module("ray-tracer_amd")
frame = importlib.import_module("ray-tracer_amd.frame")
for name in sys.argv[1:]:
    scene = rt.Scene("scenes/%s.json" % name)
    tracer = rt.Tracer(scene, device=0)
    d = frame.derive(rt, scene)
    p = tracer.collect_probe_positions()
    cell = d["keep_radius"]
    k = np.floor(p / cell).astype(np.int64)
    key = ((k[:, 0] + (1 << 20)) << 42) | ((k[:, 1] + (1 << 20)) << 21) | (k[:, 2] + (1 << 20))
    u, c = np.unique(key, return_counts=True)
    print(name, "probes", p.shape[0], "keep radius", cell, "cells", u.shape[0], "probes/cell mean %.1f median %d p90 %d max %d" % (
        c.mean(), np.median(c), np.percentile(c, 90), c.max()), flush=True)
    near = np.unique((u[:, None] + np.array([(dx << 42) + (dy << 21) + dz for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)], dtype=np.int64)[None, :]).ravel())
    zrun = np.unique((u[:, None] + np.array([-1, 0, 1], dtype=np.int64)[None, :]).ravel())
    print(name, "cells in the 27-neighbourhoods", near.shape[0], "(%.1f per occupied cell);" % (near.shape[0] / u.shape[0]),
          "in the z-neighbourhoods", zrun.shape[0], "; extent of the cell box", (k.max(axis=0) - k.min(axis=0) + 1).tolist(), flush=True)
