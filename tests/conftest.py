import importlib
import json
import os
import sys

import numpy as np
import pytest

TESTS = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(TESTS)
for p in (REPO, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(TESTS, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    """The ray-tracer_amd package (hyphenated directory, so imported by name)."""
    return importlib.import_module("ray-tracer_amd")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]

    return load


def open_scene(rt, entry):
    """Load a scene and apply a manifest entry's overrides."""
    scene = rt.Scene(os.path.join(REPO, entry["scene"]))
    s = scene.settings
    if "photons_per_light" in entry:
        s.photons_per_light = entry["photons_per_light"]
    if "fetch_size" in entry:
        s.fetch_size = entry["fetch_size"]
    if "seed" in entry and "seeded" not in entry:
        s.seeded = 1
    for key in ("use_probe_gather", "bounce_threshold", "seed", "seeded", "deterministic", "probe_sub_sample",
                "camera_time_samples", "shutter_time", "frame_time", "probe_keep_radius_scale", "bounce_store_capacity"):
        if key in entry:
            setattr(s, key, entry[key])
    if "image" in entry:  # every camera shrunk to image x image pixels
        s.image_width = s.image_height = entry["image"]
        for c in range(scene.desc.camera_count):
            scene.desc.cameras[c].width = scene.desc.cameras[c].height = entry["image"]
    return scene


@pytest.fixture(scope="session")
def scene_opener(rt):
    return lambda entry: open_scene(rt, entry)


def bits(a):
    """View a float array as unsigned integers for bit-exact comparison."""
    a = np.ascontiguousarray(a)
    return a.view({4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


def assert_bits_equal(got, expected, what=""):
    got, expected = np.asarray(got), np.asarray(expected)
    assert got.shape == expected.shape, f"{what}: shape {got.shape} vs {expected.shape}"
    if got.dtype.kind == "f":
        bad = bits(got) != bits(expected)
    else:
        bad = got != expected
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} values differ (first at {np.argwhere(bad)[0]})"


def assert_one_more_frame(frame, references):
    """`frame` is statistically one more of `references` (linear float buffers [h, w, 3] of one scene at one
    photon count, different seeds):
      * relative RMSE against each of them <= 1.25 x the largest among them,
      * total energy within four times their spread around their mean (+ 0.1 %),
      * on a 4 x 4 grid over the lit part of the image, every cell's mean within four times their largest
        deviation from their mean in that cell (+ 1 % of the brightest cell)."""
    image_check = importlib.import_module("ray-tracer_amd.image_check")
    n = len(references)
    spread = max(image_check.rel_rmse(references[i], references[j]) for i in range(n) for j in range(n) if i != j)
    worst = max(image_check.rel_rmse(frame, r) for r in references)
    assert worst <= 1.25 * spread, (worst, spread)
    energies = np.array([r.sum() for r in references])
    assert abs(frame.sum() - energies.mean()) <= 4.0 * np.abs(energies - energies.mean()).max() + 1e-3 * energies.mean(), \
        (frame.sum(), energies)
    box = image_check.lit_box(references)
    cells = np.array([image_check.regional_means(r, box) for r in references])
    centre, deviation = cells.mean(axis=0), np.abs(cells - cells.mean(axis=0)).max(axis=0)
    mine = image_check.regional_means(frame, box)
    assert (centre > 0).sum() >= 8
    excess = np.abs(mine - centre) - (4.0 * deviation + 0.01 * centre.max())
    assert (excess <= 0).all(), (mine, centre, deviation)
    return worst, spread
