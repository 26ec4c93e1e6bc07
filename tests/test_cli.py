"""The command line (`ray-tracer <scene.json>`, reference src/main.cpp:22-154) end to end on
the GPU: scene JSON -> C++ loader -> C-ABI photon pass and gather -> tonemap -> PNG16."""
import json
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import importlib

import oracle_lib as O
from conftest import REPO

image_check = importlib.import_module("ray-tracer_amd.image_check")

CLI = os.path.join(REPO, "ray-tracer_amd", "host", "ray-tracer")


def read_png16(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    at, idat, header, texts = 8, b"", None, {}
    while at < len(data):
        (n,), kind = struct.unpack(">I", data[at:at + 4]), data[at + 4:at + 8]
        body = data[at + 8:at + 8 + n]
        (crc,) = struct.unpack(">I", data[at + 8 + n:at + 12 + n])
        assert crc == zlib.crc32(kind + body), kind
        if kind == b"IHDR":
            header = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat += body
        elif kind == b"tEXt":
            key, _, value = body.partition(b"\0")
            texts[key.decode()] = value.decode()
        at += 12 + n
    w, h, depth, colour, _, _, interlace = header
    assert (depth, colour, interlace) == (16, 2, 0)
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + w * 6)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].copy().view(">u2").reshape(h, w, 3).astype(np.uint16), texts


def run_cli(*args, env=None):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=300, env=env)


def test_cli_argument_errors_exit_1(tmp_path):
    assert os.path.exists(CLI), "build the CLI with __graft_entry__.build()"
    assert run_cli().returncode == 1
    assert run_cli("a.json", "b.json").returncode == 1
    r = run_cli(str(tmp_path / "missing.json"))
    assert r.returncode == 1 and "Provided path is invalid" in r.stdout
    bad = tmp_path / "bad.json"
    bad.write_text("{ this is not json")
    r = run_cli(str(bad))
    assert r.returncode == 1 and "ERROR:" in r.stdout
    # a scene that names an unknown material fails in the loader, before any GPU work
    scene = json.load(open(os.path.join(REPO, "tests", "scenes", "furnace.json")))
    scene["$scene"]["Box"]["BackZ"]["$material"] = "NoSuchMaterial"
    broken = tmp_path / "broken.json"
    broken.write_text(json.dumps(scene))
    r = run_cli(str(broken))
    assert r.returncode == 1 and "ERROR:" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("probe_gather", [True, False])
def test_cli_renders_frame_equal_to_oracle(rt, tmp_path, probe_gather):
    scene = json.load(open(os.path.join(REPO, "tests", "scenes", "mixed.json")))
    cfg = scene["$renderConfiguration"]
    cfg.update({"$photonsPerLight": 3000, "$renderPath": str(tmp_path / "out"), "$renderName": "mixed",
                "$probeGather": probe_gather, "$probeKeepRadiusScale": 2.0})
    path = tmp_path / "scene.json"
    path.write_text(json.dumps(scene))
    env = dict(os.environ, RT_DUMP_BUFFERS=str(tmp_path / "dump"))
    r = run_cli(str(path), env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Goodbye!" in r.stdout
    image, texts = read_png16(tmp_path / "out" / "mixed.0.png")
    assert texts.get("Title") == "mixed"

    # the same frame from the oracle (counter draws, portable math)
    loaded = rt.Scene(str(path))
    orc = O.Oracle(loaded)
    desc = loaded.desc
    if probe_gather:
        setup = O.frame_setup(orc, O.RNG_COUNTER, O.MATH_PORTABLE)
        keep = setup["keep_radius"]
        emitters, _ = orc.deposit_emitters(setup["probes"], keep, keep, setup["emitter_spacing"])
        ref = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, sink_mode=O.SINK_PROBE, threads=4,
                        probes=setup["probes"], probe_cell_size=keep, keep_radius=keep,
                        store_capacity=setup["store_capacity"], preload=emitters)
        buffers = []
        for c in range(desc.camera_count):
            cam = desc.cameras[c]
            if cam.bounce_filter >= 0 or cam.light_filter >= 0:
                buffers.append(np.zeros((cam.height, cam.width, 3), dtype=np.float32))
            else:
                buffers.append(orc.gather(c, ref["store"], setup["gather_cell"], setup["min_gather_radius"])[0])
    else:
        d = orc.derive()
        ref = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, sink_mode=O.SINK_SPLAT, threads=4,
                        min_splat_radius=d["min_splat_radius"], density_cell_size=d["cell_size"])
        buffers = ref["splat"]
    for c, want in enumerate(buffers):
        cam = desc.cameras[c]
        got = np.fromfile(tmp_path / "dump" / f"camera{c}.f32", dtype=np.float32).reshape(cam.height, cam.width, 3)
        assert np.abs(want).max() > 0
        # float sums in a different order: tolerance stated relative to the brightest pixel
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-6 * float(np.abs(want).max()))
        if c == 0:
            # the PNG holds exactly the reference tonemap of the buffer the CLI produced
            iso, shutter, f_number = cam.iso, cam.shutter_time, cam.f_number
            saturation = (f_number * f_number * 12.5) / (shutter * iso) if shutter * iso > 0 else 0.0
            assert np.array_equal(image, O.tonemap(got, saturation))


@pytest.mark.gpu
@pytest.mark.parametrize("probe_gather", [True, False])
def test_cli_frame_through_the_collective_path(tmp_path, probe_gather):
    """RT_AMD_COLLECTIVE=1 takes the several-GPU path of renderFrame with the one GPU there is: an RCCL
    communicator, the gather of every camera before the all-reduce of buffers and counters.  With one
    rank the frame must be the frame of the plain path: same counters in the report, same buffers."""
    scene = json.load(open(os.path.join(REPO, "tests", "scenes", "mixed.json")))
    scene["$renderConfiguration"].update({"$photonsPerLight": 3000, "$renderName": "mixed", "$probeGather": probe_gather,
                                          "$probeKeepRadiusScale": 2.0})
    outputs = {}
    for mode in ("plain", "collective"):
        scene["$renderConfiguration"]["$renderPath"] = str(tmp_path / mode)
        path = tmp_path / f"{mode}.json"
        path.write_text(json.dumps(scene))
        env = dict(os.environ, RT_DUMP_BUFFERS=str(tmp_path / mode / "dump"))
        if mode == "collective":
            env["RT_AMD_COLLECTIVE"] = "1"
        r = run_cli(str(path), env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        report = [line for line in r.stdout.splitlines() if line.startswith(("Photon pass:", "Probe gather: kept"))]
        buffers = sorted((tmp_path / mode / "dump").glob("camera*.f32"))
        assert report and buffers
        outputs[mode] = ([line.split(" seconds=")[0] for line in report], [np.fromfile(b, dtype=np.float32) for b in buffers])
    assert outputs["plain"][0] == outputs["collective"][0]
    for a, b in zip(outputs["plain"][1], outputs["collective"][1]):
        assert a.any()
        # float sums in the order of the run (the store's append order, the splat sink's atomics)
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=1e-6 * float(np.abs(a).max()))


@pytest.mark.gpu
def test_cli_exits_1_when_a_rank_of_the_collective_path_fails(tmp_path):
    """A rank that fails before the all-reduce ends the frame: exit 1 with the rank's message, as the
    reference's main does with a worker's exception (reference src/Renderer.cpp:511-523,584-595,
    src/main.cpp:141-149), not a wait inside the collective."""
    scene = json.load(open(os.path.join(REPO, "tests", "scenes", "mixed.json")))
    scene["$renderConfiguration"].update({"$photonsPerLight": 2000, "$renderName": "mixed",
                                          "$renderPath": str(tmp_path / "out")})
    path = tmp_path / "scene.json"
    path.write_text(json.dumps(scene))
    r = run_cli(str(path), env=dict(os.environ, RT_AMD_COLLECTIVE="1", RT_AMD_FAIL_RANK="0"))
    assert r.returncode == 1
    assert "injected failure of rank 0" in r.stdout + r.stderr
    assert not os.path.exists(tmp_path / "out" / "mixed.0.png")


@pytest.mark.gpu
def test_cli_renders_keyframed_frames(rt, tmp_path):
    """$animation end to end: two frames of the keyframed scene (frame / frameRate is the frame
    time, reference src/main.cpp:95-100), each equal to the oracle's frame at that time, and
    different from one another."""
    scene = json.load(open(os.path.join(REPO, "tests", "scenes", "animated.json")))
    scene["$renderConfiguration"].update({"$photonsPerLight": 2500, "$renderPath": str(tmp_path / "out"),
                                          "$renderName": "anim", "$frameRate": 10.0, "$frameOffset": 0.0})
    frames = {}
    for frame in (3, 7):
        scene["$renderConfiguration"].update({"$startFrame": frame, "$endFrame": frame})
        path = tmp_path / f"scene{frame}.json"
        path.write_text(json.dumps(scene))
        dump = tmp_path / f"dump{frame}"
        r = run_cli(str(path), env=dict(os.environ, RT_DUMP_BUFFERS=str(dump)))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert os.path.exists(tmp_path / "out" / f"anim.{frame}.png")

        loaded = rt.Scene(str(path))
        loaded.settings.frame_time = frame / 10.0
        orc = O.Oracle(loaded)
        setup = O.frame_setup(orc, O.RNG_COUNTER, O.MATH_PORTABLE)
        keep = setup["keep_radius"]
        emitters, _ = orc.deposit_emitters(setup["probes"], keep, keep, setup["emitter_spacing"])
        ref = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, sink_mode=O.SINK_PROBE, threads=4,
                        probes=setup["probes"], probe_cell_size=keep, keep_radius=keep,
                        store_capacity=setup["store_capacity"], preload=emitters)
        want, _ = orc.gather(0, ref["store"], setup["gather_cell"], setup["min_gather_radius"])
        cam = loaded.desc.cameras[0]
        got = np.fromfile(dump / "camera0.f32", dtype=np.float32).reshape(cam.height, cam.width, 3)
        assert np.abs(want).max() > 0
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-6 * float(np.abs(want).max()))
        frames[frame] = got
    assert image_check.rel_rmse(frames[3], frames[7]) > 1e-2


@pytest.mark.gpu
def test_cornell_box_image_equivalence(rt, tmp_path):
    """Image equivalence at equal photon count on a BASELINE configuration (CornellBox.json,
    reduced to 128 x 128 and 400 k photons so that the CPU oracle finishes in seconds): the
    device frame and the oracle frame trace the same photons, so the linear buffers differ
    only by float summation order.  Stated tolerances: relative RMSE <= 1e-5; the three
    numbers of the reference's noise_metric.py equal to 1e-4 relative."""
    scene = json.load(open(os.path.join(REPO, "scenes", "CornellBox.json")))
    scene["$renderConfiguration"].update({"$width": 128, "$height": 128, "$photonsPerLight": 400000, "$seed": 5,
                                          "$renderPath": str(tmp_path / "out"), "$renderName": "cornell"})
    path = tmp_path / "cornell.json"
    path.write_text(json.dumps(scene))
    r = run_cli(str(path), env=dict(os.environ, RT_DUMP_BUFFERS=str(tmp_path / "dump")))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = np.fromfile(tmp_path / "dump" / "camera0.f32", dtype=np.float32).reshape(128, 128, 3)
    image = image_check.read_png16(tmp_path / "out" / "cornell.0.png")

    loaded = rt.Scene(str(path))
    orc = O.Oracle(loaded)
    setup = O.frame_setup(orc, O.RNG_COUNTER, O.MATH_PORTABLE)
    keep = setup["keep_radius"]
    emitters, _ = orc.deposit_emitters(setup["probes"], keep, keep, setup["emitter_spacing"])
    ref = orc.trace(rng_mode=O.RNG_COUNTER, math_mode=O.MATH_PORTABLE, sink_mode=O.SINK_PROBE, threads=8,
                    probes=setup["probes"], probe_cell_size=keep, keep_radius=keep,
                    store_capacity=setup["store_capacity"], preload=emitters)
    want, _ = orc.gather(0, ref["store"], setup["gather_cell"], setup["min_gather_radius"])
    assert float(want.mean()) > 0
    assert image_check.rel_rmse(got, want) <= 1e-5
    cam = loaded.desc.cameras[0]
    saturation = (cam.f_number ** 2 * 12.5) / (cam.shutter_time * cam.iso)
    metric_got = image_check.noise_metric(image)
    metric_want = image_check.noise_metric(O.tonemap(want, saturation))
    assert metric_want[0] > 0
    np.testing.assert_allclose(metric_got, metric_want, rtol=1e-4)
