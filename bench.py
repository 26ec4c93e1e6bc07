"""Photon-transport benchmark: photon-bounces / s on MirrorBombardment-200M.

    python bench.py --gpus N --steps K --warmup W [--scene S] [--photons P] [--sink scene|probe|splat|none]
                    [--scaling strong|weak] [--step frame|pass]

One process per GPU.  Launched through torch.distributed.run the ranks read RANK / LOCAL_RANK /
WORLD_SIZE from the environment; launched bare with --gpus N > 1 this process starts the N ranks
itself (before anything touches a GPU) and relays rank 0's line.  --gpus must equal WORLD_SIZE.

A step is one frame of the scene after the camera-side set-up (scene, trees and — in the scene's
default probe-gather mode — the probe pass and its keep-test index are resident in HBM before the
timed region; their device time is reported as `setup_ms`, and --step frame+probes puts them inside
the step): sinks reset, emitter deposits (rank 0 only: they enter the sum once), the photon pass
over this rank's share of the photon indices, the gather of this rank's deposits into its camera
buffers, then one RCCL all-reduce(sum) per camera buffer and one of the counters when N > 1.
Rank r of N traces photon indices [r*P/N, (r+1)*P/N) of every light; P is the scene's budget
(--scaling strong, the default: MirrorBombardment-200M split N ways) or N times it (--scaling weak).
--step pass times the photon pass alone (reset, pass, all-reduce), without the gather.
The metric counts ray casts that produced a valid hit (photon-bounces) over whole-step wall time.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel, traceKernel.  Its duration is
measured live with HIP events the library records on the tracer's stream around that kernel
(rt_read_timings).  The kernel is bound by double-precision vector arithmetic (individually rounded
operations, two waves per SIMD), not by HBM: `achieved` is the path's algorithmic FP64 operations —
per cast and volume 150 for the transforms, 30 per slab test at two per level of a balanced tree, 90 per
triangle test at two per mesh (algorithmic_flops_per_cast) — times the casts of a launch, over the live
kernel time, against the vector units' FP64 rate without fused multiply-adds.  `valu_busy` is the share
of all SIMD cycles that issued a vector instruction (SQ_ACTIVE_INST_VALU of the committed rocprofv3
record of exactly this workload, profiles/counters.json) and `traffic` the HBM bytes of one launch from
the FETCH_SIZE / WRITE_SIZE passes of the same record.  The HBM-priced figure of the path's algorithmic
bytes is kept under `algorithmic`: for a scene that lives in cache it exceeds the HBM peak and is no roofline.
"""
import argparse
import importlib
import json
import math
import os
import socket
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

SCENE = os.path.join(REPO, "scenes", "MirrorBombardment-200M.json")
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
SIMDS, CLOCK_GHZ = 1024, 2.4   # 256 CUs x 4 SIMDs, peak shader clock
# FP64 vector peak of the MI355X data sheet, 78.6 TFLOP/s, counts a fused multiply-add as two operations.
# This path may not fuse (its products are rounded one by one, as the reference's are: -ffp-contract=off),
# so what the vector units can deliver here is one operation per lane and cycle: half of that.
FP64_VECTOR_PEAK_TFLOPS = 78.6 / 2
COUNTER_NAMES = ("photons", "casts", "bounces", "deposits", "kept", "culled", "store_attempted", "store_size",
                 "store_dropped", "splat_total", "density_dropped")


def algorithmic_bytes_per_cast(desc):
    """Compulsory bytes one scene cast touches if nothing were cached: per mesh volume two visited nodes
    per tree level (48 B box + 8 B links) and two tested triangles (72 B of positions), plus the winning
    triangle's 72 B of normals once per cast."""
    total = 72.0
    for v in range(desc.volume_count):
        vol = desc.volumes[v]
        if vol.type == 0:
            tris = max(1, desc.meshes[vol.mesh].triangle_count)
            depth = max(1, math.ceil(math.log2(tris)))
            total += 2 * depth * 56 + 2 * 72
    return total


def algorithmic_flops_per_cast(desc):
    """Double-precision operations one scene cast needs by the path's own arithmetic, whatever the tree
    walked: per volume the ray's transform into object space and the hit's way back (150), per mesh volume
    two slab tests per level of a balanced tree over its triangles (30 each) and two watertight triangle
    tests (90 each); an analytic volume (sphere, quad, plane) one intersection (40)."""
    total = 0.0
    for v in range(desc.volume_count):
        vol = desc.volumes[v]
        total += 150.0
        if vol.type == 0:
            tris = max(1, desc.meshes[vol.mesh].triangle_count)
            depth = max(1, math.ceil(math.log2(tris)))
            total += 2 * depth * 30 + 2 * 90
        else:
            total += 40.0
    return total


def cpu_baseline(rt, scene_path, sink, setup, photons):
    """The CPU path timed on a bounded sample of the same workload (same scene, `photons` photons per
    light instead of the full budget; cost is linear in photons), on this machine's host cores.

    Preferred: the REFERENCE itself (oracle/_ref/refdump bench: its own multithreaded Worker loop,
    photonPassSeconds as it reports it); its bounce count comes from the oracle run at the same photon
    count (the reference does not count delta bounces).  Fallback: the oracle port."""
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
    refdump = os.path.join(REPO, "oracle", "_ref", "refdump")
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
                              f"{ref['photon_pass_seconds']:.2f} (includes its single-threaded probe pass, excludes "
                              f"its gather); {r['bounces']} bounces counted by the oracle at the same photon count",
                    "port_value": r["bounces"] / port_seconds,
                }
        except Exception as e:  # the reference binary is optional on the GPU box
            result["reference_error"] = str(e)[:200]
    return result


def committed_counters(workload, sink, photons_per_launch):
    """The rocprofv3 PMC record of one traceKernel launch of exactly this workload, or None."""
    try:
        with open(os.path.join(REPO, "profiles", "counters.json")) as f:
            for entry in json.load(f)["launches"]:
                if (entry["workload"], entry["sink"], entry["photons_per_launch"]) == (workload, sink, photons_per_launch):
                    return entry
    except (OSError, KeyError, ValueError):
        pass
    return None


def launch_ranks(args_list, gpus):
    """Start `gpus` ranks of this script under torch.distributed.run and relay their output."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + args_list
    return subprocess.run(cmd).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--photons", type=int, default=0, help="photons per light (default: the scene's 200 M)")
    ap.add_argument("--sink", default="scene", choices=["scene", "probe", "splat", "none"])
    ap.add_argument("--scene", default=SCENE)
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: the scene's photon budget is split across the GPUs (the metric's workload); "
                         "weak: every GPU traces the scene's full budget (global budget = N x scene)")
    ap.add_argument("--step", default="frame", choices=["frame", "pass", "frame+probes"],
                    help="frame: emitter deposits, photon pass, gather, all-reduce; pass: the photon pass and the all-reduce; "
                         "frame+probes: the frame with its camera-side set-up inside the step as well (probe pass of every "
                         "camera, probe index and store), which the reference's photonPassSeconds includes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reuse-order", action="store_true",
                    help="steps after the first reuse the pass's emission order (the frames of a sequence: same ids, "
                         "same seed, same order); by default every step works it out anew, sort included")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend (nccl = RCCL; gloo only to rehearse the multi-rank path on one GPU)")
    ap.add_argument("--collective", default="auto", choices=["auto", "always"],
                    help="always: set up the process group and run the all-reduces with one rank too (exercises the RCCL path on one GPU)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal: every rank uses GPU 0 (with --backend gloo)")
    ap.add_argument("--store-share", type=int, default=0,
                    help="rehearsal on one GPU of one rank's step among this many: with --photons <share> the store is held "
                         "to the part of it a rank among N keeps")
    ap.add_argument("--cpu-photons", type=int, default=40_000_000,
                    help="photons per light of the CPU baseline sample (about 20 s of CPU work)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            # no launcher: start the ranks from here, while this process has not touched a GPU
            raise SystemExit(launch_ranks(sys.argv[1:], args.gpus))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={os.environ['WORLD_SIZE']}: "
                         "launch one rank per GPU and pass their number as --gpus")

    # Rank 0's JSON line is the only thing this process writes to its standard output: libraries that
    # print there (RCCL's version banner when a communicator is made) are sent to standard error.
    result_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)

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
    collective = world > 1 or args.collective == "always"
    if collective:
        kwargs = {}
        if "MASTER_ADDR" not in os.environ:  # one rank started without a launcher
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                kwargs = dict(init_method=f"tcp://127.0.0.1:{s.getsockname()[1]}", rank=rank, world_size=world)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device, **kwargs)
        else:
            dist.init_process_group("gloo", **kwargs)

    rt = importlib.import_module("ray-tracer_amd")
    parallel = importlib.import_module("ray-tracer_amd.parallel")
    frame = importlib.import_module("ray-tracer_amd.frame")

    scene = rt.Scene(args.scene)
    if args.photons:
        scene.settings.photons_per_light = args.photons
    if args.scaling == "weak":
        # per-photon flux is Phi / (global count), so the image keeps its brightness
        scene.settings.photons_per_light = int(scene.settings.photons_per_light) * world
    photons = int(scene.settings.photons_per_light)
    lights = int(scene.desc.light_count)
    sink = args.sink
    if sink == "scene":
        sink = "probe" if scene.settings.use_probe_gather else "splat"
    workload = os.path.basename(args.scene).replace(".json", "")

    tracer = rt.Tracer(scene, device=local_rank)
    stream = torch.cuda.current_stream(device)
    tracer.set_stream(stream.cuda_stream)
    setup = frame.prepare_sinks(rt, scene, tracer, sink)  # derived radii, probe pass, keep-test index, store sizing
    gather_cell = setup["scene_depth_footprint"] if setup["scene_depth_footprint"] > 0.0 else setup["cell_size"]
    gathering = [c for c in range(scene.desc.camera_count)
                 if scene.desc.cameras[c].bounce_filter < 0 and scene.desc.cameras[c].light_filter < 0]

    # camera buffers live in torch tensors so that RCCL can reduce them in place
    buffers = []
    for cam, (w, h) in enumerate(tracer.cameras):
        t = torch.zeros((h, w, 3), dtype=torch.float32, device=device)
        tracer.bind_camera_buffer(cam, t.data_ptr())
        buffers.append(t)

    first, count = parallel.shard_range(photons, rank, world)
    store_ranks = world if world > 1 else max(1, args.store_share)
    sink_mode = {"none": rt.tracer.SINK_NONE, "probe": rt.tracer.SINK_PROBE, "splat": rt.tracer.SINK_SPLAT}[sink]
    whole_frame = args.step in ("frame", "frame+probes")
    with_probes = args.step == "frame+probes" and sink == "probe"

    def camera_side_setup():
        """The frame's set-up as Renderer::renderFrame runs it, on the device: the probe pass of every gathering
        camera, the keep-test index over the records and the store behind it (reference src/Renderer.cpp:297-419),
        and this rank's share of the store when the frame is spread over ranks."""
        tracer.forget_gather_points()
        for c in gathering:
            tracer.count_gather_points(c)
        tracer.set_probes_from_cameras(setup["keep_radius"], setup["keep_radius"], 256, int(scene.settings.bounce_store_capacity))
        if store_ranks > 1:
            # the frame has ONE store: a rank keeps the emitter deposits (rank 0) plus 1/N of the rest
            tracer.reset()
            frame.share_store(tracer, setup["store_capacity"], max(gather_cell * 0.5, 1e-6), rank, store_ranks)

    setup_ms = None
    if sink == "probe":
        camera_side_setup()
        setup_ms = tracer.setup_timings()

    kernel_ms, sort_ms, gather_ms = [], [], []

    def step(timed):
        if with_probes:
            camera_side_setup()
            if os.environ.get("RT_BENCH_VERBOSE"):
                tracer.synchronize()
                print(f"[bench] camera-side set-up done at {time.perf_counter():.4f}", file=sys.stderr)
        tracer.reset()
        if whole_frame and sink == "probe" and rank == 0:
            tracer.deposit_emitters(max(gather_cell * 0.5, 1e-6))
        if count > 0:  # an empty share traces nothing (a count of 0 would mean "all the rest")
            tracer.trace(sink_mode=sink_mode, first_photon=first, photon_count=count, reuse_order=args.reuse_order)
        if whole_frame and sink == "probe":
            if timed:
                tracer.synchronize()  # the gather waits for the pass anyway: this only separates the two times
            t0 = time.perf_counter()
            for c in gathering:
                tracer.gather(c, gather_cell, setup["min_splat_radius"])
            if timed:
                gather_ms.append(1e3 * (time.perf_counter() - t0))
        if collective:
            for t in buffers:
                dist.all_reduce(t)
        c = tracer.counters()  # synchronises the stream
        if os.environ.get("RT_BENCH_VERBOSE"):
            print(f"[bench] step done at {time.perf_counter():.4f}", file=sys.stderr)
        totals = torch.tensor([c[k] for k in COUNTER_NAMES], dtype=torch.int64, device=device)
        if collective:
            dist.all_reduce(totals)
        if timed and count > 0:
            s_ms, t_ms = tracer.timings()
            sort_ms.append(s_ms)
            kernel_ms.append(t_ms)
        return dict(zip(COUNTER_NAMES, [int(v) for v in totals.tolist()]))

    # RT_BENCH_FAIL_RANK=<rank>: that rank dies before its first step (a test switch: the job must end with a
    # failure promptly, not wait for the dead rank in a collective)
    if os.environ.get("RT_BENCH_FAIL_RANK") is not None and int(os.environ["RT_BENCH_FAIL_RANK"]) == rank:
        raise SystemExit(f"rank {rank}: injected failure (RT_BENCH_FAIL_RANK)")

    def fence():
        torch.cuda.synchronize(device)
        if collective:
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
    if collective:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / max(1, args.steps)
        bounces = totals["bounces"]
        value = bounces / (ms_per_step * 1e-3)
        desc = scene.desc
        local = tracer.counters()
        avg_kernel_s = 1e-3 * sum(kernel_ms) / max(1, len(kernel_ms))
        avg_sort_ms = sum(sort_ms) / max(1, len(sort_ms))
        pmc = committed_counters(workload, sink, count * lights)

        # Roofline of the trace kernel: the path's algorithmic double-precision operations over the kernel's
        # measured time, against the vector units' non-fused FP64 rate.  (The share of SIMD cycles that issue
        # a vector instruction, from the committed counter record of this workload, is kept as `valu_busy`: a
        # utilisation, not a bound — a kernel that issued more instructions per cast would score higher.)
        flops_per_cast = algorithmic_flops_per_cast(desc)
        launch_casts = local["casts"] + (local["splat_total"] if sink == "splat" else 0)  # a splat casts a shadow ray
        roofline = {"bound": "fp64-vector", "achieved": None, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None,
                    "traffic": None, "kernel": "traceKernel", "kernel_ms": 1e3 * avg_kernel_s, "sort_kernels_ms": avg_sort_ms,
                    "flops_per_cast": flops_per_cast, "casts_per_launch": launch_casts, "valu_busy": None, "counters": None}
        hbm = None
        if avg_kernel_s > 0:
            achieved = flops_per_cast * launch_casts / avg_kernel_s / 1e12
            roofline.update(achieved=achieved, frac=achieved / FP64_VECTOR_PEAK_TFLOPS)
        if pmc is not None and avg_kernel_s > 0:
            busy_cycles = 4.0 * pmc["SQ_ACTIVE_INST_VALU"]  # the counter ticks once per four cycles
            roofline.update(valu_busy=busy_cycles / avg_kernel_s / 1e9 / (SIMDS * CLOCK_GHZ), traffic=pmc["hbm_bytes"],
                            counters=pmc["source"],
                            valu_instructions_per_cast=pmc["SQ_INSTS_VALU"] / max(1, local["casts"]),
                            lanes_active=pmc.get("lanes_active"), l2_hit_rate=pmc.get("l2_hit_rate"),
                            l2_request_bytes=pmc.get("l2_request_bytes"), wait_memory_frac=pmc.get("wait_memory_frac"))
            hbm_rate = pmc["hbm_bytes"] / avg_kernel_s / 1e9
            hbm = {"bound": "hbm", "achieved": hbm_rate, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_rate / HBM_PEAK_GBS,
                   "note": "measured HBM traffic of the trace kernel: the scene lives in cache, HBM is not the limiter"}
        # the path priced against HBM as if nothing were cached (no roofline: see the module docstring)
        per_cast = algorithmic_bytes_per_cast(desc)
        casts = local["casts"] + (local["splat_total"] if sink == "splat" else 0)
        sink_bytes = 0.0
        if sink == "probe":
            sink_bytes = local["deposits"] * 27 * 16 + local["kept"] * 52
        elif sink == "splat":
            sink_bytes = local["splat_total"] * 12.0
        algorithmic = casts * per_cast + sink_bytes
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
                "workload": workload,
                "photons_per_light": photons,
                "lights": lights,
                "sink": sink,
                "scene_sink": "probe" if scene.settings.use_probe_gather else "splat",
                "emission_order": "reused from the step before" if args.reuse_order else "sorted anew every step",
                "step": ("probe pass + probe index + " if with_probes else "")
                        + ("emitter deposits + photon pass + gather" if whole_frame and sink == "probe" else "photon pass")
                        + (" + all-reduce" if collective else ""),
                "bounce_threshold": int(scene.settings.bounce_threshold),
                "meshes": "procedural stand-ins (reference OBJ files are Git-LFS pointers)",
                "parallelism": f"photon-range x{world}",
                "probes": setup.get("probe_count", 0),
            },
            "counters": totals,
            "photons_per_s": totals["photons"] / (ms_per_step * 1e-3),
            "casts_per_s": totals["casts"] / (ms_per_step * 1e-3),
            "stages_ms": {"sort": avg_sort_ms, "trace": 1e3 * avg_kernel_s,
                          "gather": (sum(gather_ms) / len(gather_ms)) if gather_ms else 0.0},
            # device time of the camera-side set-up of a frame (outside the step unless --step frame+probes)
            "setup_ms": setup_ms,
            "photon_pass_bounces_per_s": local["bounces"] * world / ((avg_sort_ms * 1e-3) + avg_kernel_s)
            if avg_kernel_s > 0 else None,
            "roofline": roofline,
            "algorithmic": {"bytes": algorithmic, "bytes_per_cast": per_cast,
                            "rate_if_from_hbm_GBs": algorithmic / avg_kernel_s / 1e9 if avg_kernel_s > 0 else None},
        }
        if hbm is not None:
            out["roofline_hbm"] = hbm
        for key in ("frac",):
            assert roofline[key] is None or 0.0 < roofline[key] <= 1.0, roofline
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(rt, args.scene, sink, setup, args.cpu_photons)
        print(json.dumps(out), file=result_out, flush=True)

    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
