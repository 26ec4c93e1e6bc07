"""The host classes' Material virtuals (reference include/Material.h:56-101) evaluate the device's BSDFs:
same results as the function-level entry points with the draws a std::mt19937 generator gives, and the
reference's number of draws per sample (MI355X only: there is no host BSDF code)."""
import os
import random

import numpy as np
import pytest

from conftest import REPO, assert_bits_equal

pytestmark = pytest.mark.gpu


def mt_draws(seed, n):
    """uniform_real_distribution<double>(0, 1) over std::mt19937 (libstdc++ generate_canonical: two
    32-bit outputs per double) — what the reference's RandomGenerator::value() returns."""
    legacy = np.random.RandomState(seed)  # same 32-bit Mersenne Twister, init_genrand seeding
    raw = legacy.randint(0, 1 << 32, size=2 * n, dtype=np.uint64)
    lo, hi = raw[0::2].astype(np.float64), raw[1::2].astype(np.float64)
    return (lo + hi * 4294967296.0) / 18446744073709551616.0


def test_material_virtuals_run_the_device_bsdfs(rt):
    scene = rt.Scene(os.path.join(REPO, "tests", "scenes", "mixed.json"))
    tracer = __import__("importlib").import_module("ray-tracer_amd.tracer")
    desc = scene.desc
    rng = random.Random(5)
    kinds = set()
    for m in range(desc.material_count):
        mat = desc.materials[m]
        kinds.add(mat.type)
        for trial in range(6):
            n = np.array([rng.gauss(0, 1) for _ in range(3)])
            n /= np.linalg.norm(n)
            i = np.array([rng.gauss(0, 1) for _ in range(3)])
            i /= np.linalg.norm(i)
            if trial % 3 and i @ n > 0:
                i = -i  # mostly arriving from the front
            seed = 100 * m + trial
            got = scene.material_sample(m, i, n, seed)
            front = (-i) @ n > 0
            draws = {0: 2, 1: 0, 2: 1, 3: 2 if front else 0}[mat.type]
            assert got[8] == draws, (mat.type, front)
            u = list(mt_draws(seed, 2)) if draws else [0.0, 0.0]
            if draws == 1:
                u[1] = 0.0
            row = np.array([[mat.type, mat.color[0], mat.color[1], mat.color[2], mat.alpha, mat.ior, *i, *n, u[0], u[1]]])
            want = tracer.rows("rt_bsdf_sample", row, 10)[0]
            assert_bits_equal(np.array(got[:8]), want[:8], f"material {m} sample")
            wo = np.array([rng.gauss(0, 1) for _ in range(3)])
            wo /= np.linalg.norm(wo)
            ev = scene.material_evaluate(m, -i, wo, n)
            erow = np.array([[mat.type, mat.color[0], mat.color[1], mat.color[2], mat.alpha, *(-i), *wo, *n]])
            want_ev = tracer.rows("rt_bsdf_evaluate", erow, 4)[0]
            assert_bits_equal(np.array(ev[:4]), want_ev, f"material {m} evaluate")
            assert ev[4] == (1.0 if mat.type in (1, 2) else 0.0)
    assert kinds == {0, 1, 2, 3}
