"""Photon-pass benchmark: photon-bounces / s on MirrorBombardment-200M.

    python bench.py --gpus N --steps K --warmup W [--photons P] [--sink probe|splat|none]

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the environment when launched
through torch.distributed.run).  A step is one photon pass over the whole photon budget
of the frame: rank r traces photon indices [r*P/N, (r+1)*P/N) of every light, where P is
the scene's budget x N by default (weak scaling: every GPU traces the scene's 200 M
photons, the frame holds N x 200 M) or the scene's budget itself with --scaling strong;
then the per-camera buffers and the counters are summed across ranks with one RCCL
all-reduce.  Inputs (scene, tree, probe grid) are resident in HBM before the timed
region.  The metric counts ray casts that produced a valid hit (a photon-bounce).

Rank 0 prints ONE JSON line.  `roofline` is that of the dominant kernel, traceKernel: achieved =
algorithmic bytes of the launch (algorithmic_bytes_per_cast x casts + sink bytes) / its duration; its
duration comes from HIP events the library records on the tracer's stream around that
kernel (rt_read_timings); `roofline.traffic` is the HBM byte count of the same launch
measured with rocprofv3 PMC passes and committed in profiles/traffic.json (null when the
run's configuration has no committed measurement).
"""
import argparse
import importlib
import json
import math
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

SCENE = os.path.join(REPO, "scenes", "MirrorBombardment-200M.json")
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def algorithmic_bytes_per_cast(desc):
    """Compulsory bytes one scene cast touches: per mesh volume two visited nodes per tree
    level (48 B box + 8 B links) and two tested triangles (72 B of positions), plus the
    winning triangle's 72 B of normals once per cast."""
    total = 72.0
    for v in range(desc.volume_count):
        vol = desc.volumes[v]
        if vol.type == 0:
            tris = max(1, desc.meshes[vol.mesh].triangle_count)
            depth = max(1, math.ceil(math.log2(tris)))
            total += 2 * depth * 56 + 2 * 72
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--photons", type=int, default=0, help="photons per light (default: the scene's 200 M)")
    ap.add_argument("--sink", default="scene", choices=["scene", "probe", "splat", "none"])
    ap.add_argument("--scene", default=SCENE)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every GPU traces the scene's full photon budget (global budget = N x scene); "
                         "strong: the scene's budget is split across the GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend (nccl = RCCL; gloo only to rehearse the multi-rank path on one GPU)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal: every rank uses GPU 0 (with --backend gloo)")
    ap.add_argument("--cpu-photons", type=int, default=40_000_000,
                    help="photons per light of the CPU baseline sample (about 20 s of CPU work)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the photon pass has no CPU fallback")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    rt = importlib.import_module("ray-tracer_amd")
    parallel = importlib.import_module("ray-tracer_amd.parallel")
    frame = importlib.import_module("ray-tracer_amd.frame")

    scene = rt.Scene(args.scene)
    if args.photons:
        scene.settings.photons_per_light = args.photons
    per_gpu = int(scene.settings.photons_per_light)
    if args.scaling == "weak":
        # per-photon flux is Phi / (global count), so the image keeps its brightness
        scene.settings.photons_per_light = per_gpu * world
    photons = int(scene.settings.photons_per_light)
    lights = int(scene.desc.light_count)
    sink = args.sink
    if sink == "scene":
        sink = "probe" if scene.settings.use_probe_gather else "splat"

    tracer = rt.Tracer(scene, device=local_rank)
    stream = torch.cuda.current_stream(device)
    tracer.set_stream(stream.cuda_stream)
    setup = frame.prepare_sinks(rt, scene, tracer, sink)  # derived radii, probe pass, store sizing

    # camera buffers live in torch tensors so that RCCL can reduce them in place
    buffers = []
    for cam, (w, h) in enumerate(tracer.cameras):
        t = torch.zeros((h, w, 3), dtype=torch.float32, device=device)
        tracer.bind_camera_buffer(cam, t.data_ptr())
        buffers.append(t)
    counter_names = ("photons", "casts", "bounces", "deposits", "kept", "culled", "store_attempted", "splat_total")

    first, count = parallel.shard_range(photons, rank, world)
    sink_mode = {"none": rt.tracer.SINK_NONE, "probe": rt.tracer.SINK_PROBE, "splat": rt.tracer.SINK_SPLAT}[sink]

    kernel_ms, sort_ms = [], []

    def step(timed):
        tracer.reset()
        tracer.trace(sink_mode=sink_mode, first_photon=first, photon_count=count)
        if world > 1:
            for t in buffers:
                dist.all_reduce(t)
        c = tracer.counters()  # synchronises the stream
        totals = torch.tensor([c[k] for k in counter_names], dtype=torch.int64, device=device)
        if world > 1:
            dist.all_reduce(totals)
        if timed:
            s_ms, t_ms = tracer.timings()
            sort_ms.append(s_ms)
            kernel_ms.append(t_ms)
        return dict(zip(counter_names, [int(v) for v in totals.tolist()]))

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    totals = None
    for _ in range(args.steps):
        totals = step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / max(1, args.steps)
        bounces = totals["bounces"]
        value = bounces / (ms_per_step * 1e-3)
        # roofline of the dominant (only) kernel of a step, per launch on this rank
        desc = scene.desc
        local = tracer.counters()
        per_cast = algorithmic_bytes_per_cast(desc)
        casts = local["casts"] + (local["splat_total"] if sink == "splat" else 0)
        sink_bytes = 0.0
        if sink == "probe":
            sink_bytes = local["deposits"] * 27 * 16 + local["kept"] * 52
        elif sink == "splat":
            sink_bytes = local["splat_total"] * 12.0 * len(buffers)
        algorithmic = casts * per_cast + sink_bytes
        avg_kernel_s = 1e-3 * sum(kernel_ms) / max(1, len(kernel_ms))
        achieved = algorithmic / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        traffic = None
        try:
            with open(os.path.join(REPO, "profiles", "traffic.json")) as f:
                for entry in json.load(f)["launches"]:
                    if (entry["workload"] == os.path.basename(args.scene).replace(".json", "")
                            and entry["sink"] == sink and entry["photons_per_launch"] == count * lights):
                        traffic = entry["hbm_bytes"]
        except (OSError, KeyError, ValueError):
            traffic = None
        out = {
            "metric": "photon-bounces/sec (whole node), MirrorBombardment-200M",
            "value": value,
            "unit": "photon-bounces/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": os.path.basename(args.scene).replace(".json", ""),
                "photons_per_light": photons,
                "lights": lights,
                "sink": sink,
                "bounce_threshold": int(scene.settings.bounce_threshold),
                "meshes": "procedural stand-ins (reference OBJ files are Git-LFS pointers)",
                "parallelism": f"photon-range x{world}",
                "probes": setup.get("probe_count", 0),
            },
            "counters": totals,
            "photons_per_s": totals["photons"] / (ms_per_step * 1e-3),
            "casts_per_s": totals["casts"] / (ms_per_step * 1e-3),
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": "traceKernel",
                "kernel_ms": 1e3 * avg_kernel_s,
                "sort_kernel_ms": sum(sort_ms) / max(1, len(sort_ms)),
                "algorithmic_bytes": algorithmic,
                "algorithmic_bytes_per_cast": per_cast,
            },
        }
        # secondary bound: FP64 vector issue.  Flops per cast as priced for this path: per mesh volume
        # 150 (ray transform) + 2 * depth * 30 (slab tests) + 2 * 90 (triangle tests); peak = 256 CUs x
        # 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 TFLOP/s for individually rounded operations (no FMA contraction)
        flops_per_cast = 0.0
        for v in range(desc.volume_count):
            vol = desc.volumes[v]
            if vol.type == 0:
                tris = max(1, desc.meshes[vol.mesh].triangle_count)
                flops_per_cast += 150 + 2 * max(1, math.ceil(math.log2(tris))) * 30 + 2 * 90
            else:
                flops_per_cast += 150 + 40
        valu = casts * flops_per_cast / avg_kernel_s / 1e12 if avg_kernel_s > 0 else 0.0
        out["roofline_valu"] = {"bound": "fp64-valu", "achieved": valu, "peak": 39.3, "unit": "TFLOP/s",
                                "frac": valu / 39.3, "flops_per_cast": flops_per_cast}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = frame.cpu_baseline(rt, args.scene, sink, setup, args.cpu_photons)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
