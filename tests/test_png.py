"""The CLI's 16-bit PNG writer (reference src/PngWriter.cpp, src/PngWriteSession.cpp:118-173) through
the host library: chunk CRCs, the zlib stream's Adler-32 and the pixels, read back with zlib."""
import ctypes as C
import importlib

import numpy as np
import pytest

from test_cli import read_png16


@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (131, 257), (700, 1100)])
def test_png16_round_trip(tmp_path, shape):
    rt = importlib.import_module("ray-tracer_amd")
    h, w = shape
    rng = np.random.default_rng(w * 1000 + h)
    rgb = rng.integers(0, 65536, size=(h, w, 3), dtype=np.uint16)
    rgb[0, 0] = (0, 65535, 1)
    path = tmp_path / "image.png"
    rc = rt.host_lib().rth_write_png16(str(path).encode(), rgb.ctypes.data_as(C.c_void_p), w, h, b"a title")
    assert rc == 0, rt.host_lib().rth_last_error().decode()
    back, texts = read_png16(path)  # checks every chunk's CRC; zlib checks the Adler-32
    assert np.array_equal(back, rgb)
    assert texts == {"Title": "a title"}


def test_png16_rejects_bad_arguments(tmp_path):
    rt = importlib.import_module("ray-tracer_amd")
    assert rt.host_lib().rth_write_png16(str(tmp_path / "x.png").encode(), None, 4, 4, b"") != 0


@pytest.mark.parametrize("shape", [(3, 5), (131, 257)])
def test_png16_opens_in_an_independent_decoder(tmp_path, shape):
    """A PNG reader that shares no code with the writer or with read_png16 (Pillow: its own chunk
    parser over zlib) accepts the file: verify() walks every chunk and its CRC, the header says 16 bits
    of RGB, the title comes back, and the pixels it hands out — Pillow keeps the high byte of a 16-bit
    RGB sample — are the high bytes written.  (Byte-for-byte equality with libpng's own output, which
    depends on its filter and deflate choices, is not claimed.)"""
    Image = pytest.importorskip("PIL.Image")
    rt = importlib.import_module("ray-tracer_amd")
    h, w = shape
    rng = np.random.default_rng(7 * w + h)
    rgb = rng.integers(0, 65536, size=(h, w, 3), dtype=np.uint16)
    path = tmp_path / "image.png"
    assert rt.host_lib().rth_write_png16(str(path).encode(), rgb.ctypes.data_as(C.c_void_p), w, h, b"a title") == 0
    with Image.open(path) as probe:
        probe.verify()
    with Image.open(path) as image:
        assert image.size == (w, h) and image.format == "PNG"
        assert image.text.get("Title") == "a title"
        pixels = np.asarray(image.convert("RGB"))
    assert pixels.shape == (h, w, 3)
    assert np.array_equal(pixels, (rgb >> 8).astype(np.uint8))
