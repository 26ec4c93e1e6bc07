"""rt_portable.h: Philox known answers and the accuracy of the shared sin/cos/acos."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO

SRC = r"""
#include "rt_portable.h"
void philox(unsigned* c, unsigned k0, unsigned k1) { rt_philox4x32_10(c, k0, k1); }
void draws(unsigned seed, unsigned light, unsigned long long photon, unsigned stage, int n, double* out)
{ rt_philox_state s; rt_philox_begin(&s, seed, light, photon, stage); for (int i = 0; i < n; ++i) out[i] = rt_philox_next(&s); }
void trig(int n, const double* x, double* s, double* c) { for (int i = 0; i < n; ++i) rt_sincos(x[i], &s[i], &c[i]); }
void arccos(int n, const double* x, double* y) { for (int i = 0; i < n; ++i) y[i] = rt_acos(x[i]); }
"""


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("portable")
    (d / "p.c").write_text(SRC)
    so = d / "p.so"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(REPO, "include"),
                    str(d / "p.c"), "-o", str(so), "-lm"], check=True)
    return C.CDLL(str(so))


def test_philox_known_answers(lib):
    # Random123 kat_vectors: philox4x32-10
    cases = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
             ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
             ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
              (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, expected in cases:
        c = (C.c_uint32 * 4)(*ctr)
        lib.philox(c, key[0], key[1])
        assert tuple(c) == expected


def test_draw_stream_is_a_pure_function_of_its_key(lib):
    def draws(seed, light, photon, stage, n=6):
        out = (C.c_double * n)()
        lib.draws(seed, light, C.c_uint64(photon), stage, n, out)
        return np.array(out)

    a = draws(5, 1, 123456789012, 3)
    assert np.array_equal(a, draws(5, 1, 123456789012, 3))
    assert ((a >= 0) & (a < 1)).all()
    for other in (draws(6, 1, 123456789012, 3), draws(5, 2, 123456789012, 3), draws(5, 1, 123456789013, 3),
                  draws(5, 1, 123456789012, 4)):
        assert not np.array_equal(a, other)
    # draw k comes from block k >> 1: a shorter request is a prefix of a longer one
    assert np.array_equal(draws(5, 1, 7, 0, 3), draws(5, 1, 7, 0, 6)[:3])


def test_trig_within_one_ulp_of_libm(lib):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(0, 2 * np.pi, 200000), rng.uniform(-0.8, 0.8, 20000),
                        np.array([0.0, np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi])])
    s, c = np.zeros_like(x), np.zeros_like(x)
    lib.trig(x.size, x.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p))
    for got, ref in ((s, np.sin(x)), (c, np.cos(x))):
        err = np.abs(got - ref) / np.maximum(np.spacing(np.abs(ref)), 1e-300)
        big = np.abs(ref) > 1e-3  # near a zero crossing an ulp of the result is tiny; check absolute there
        assert err[big].max() <= 1.0
        assert np.abs(got - ref)[~big].max() < 2e-16
    y = rng.uniform(-1, 1, 200000)
    a = np.zeros_like(y)
    lib.arccos(y.size, y.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p))
    assert (np.abs(a - np.arccos(y)) / np.spacing(np.arccos(y))).max() <= 1.0
