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
                "camera_time_samples", "shutter_time", "frame_time", "probe_keep_radius_scale"):
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
