"""Worker of test_frame_ranks.py: renders one frame on GPU 0 as rank RANK of WORLD_SIZE (gloo) and,
on rank 0, saves the reduced camera buffers and counters."""
import importlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    scene_path, photons, probe_gather, out = sys.argv[1], int(sys.argv[2]), sys.argv[3] == "1", sys.argv[4]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    rt = importlib.import_module("ray-tracer_amd")
    frame = importlib.import_module("ray-tracer_amd.frame")
    scene = rt.Scene(scene_path)
    scene.settings.photons_per_light = photons
    scene.settings.use_probe_gather = 1 if probe_gather else 0
    scene.settings.seeded, scene.settings.seed = 1, 9
    if len(sys.argv) > 5:
        scene.settings.bounce_store_capacity = int(sys.argv[5])  # a store small enough to overflow
    device = torch.device("cuda", 0)
    tracer = rt.Tracer(scene, device=0)
    buffers = []
    for cam, (w, h) in enumerate(tracer.cameras):
        t = torch.zeros((h, w, 3), dtype=torch.float32, device=device)
        tracer.bind_camera_buffer(cam, t.data_ptr())
        buffers.append(t)
    counters = frame.render_frame(rt, scene, tracer, buffers, rank, world, dist if world > 1 else None)
    if rank == 0:
        np.savez(out, **{f"camera{c}": b.cpu().numpy() for c, b in enumerate(buffers)})
        json.dump({k: (v if isinstance(v, int) else None) for k, v in counters.items()}, open(out + ".json", "w"))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
