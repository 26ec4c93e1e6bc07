"""The control flow of a frame spread over several GPUs of one process (host/RankFrames.h, used by
Renderer::renderFrame under RT_AMD_DEVICES), with stand-in shares and a stand-in collective — no GPU.

The reference rethrows a worker's exception after stopping the other workers and has a stall guard
(reference src/Renderer.cpp:511-523,584-595); the sharded frame must likewise end with the failing
rank's message and must never enter its collective with a rank missing.
"""
import ctypes as C
import time

import pytest


def run_ranks(rt, world, failing_rank=-1, cancel_after_pieces=-1):
    lib = rt.host_lib()
    lib.rth_test_rank_frames.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                         C.c_char_p, C.c_int32]
    entered, traced = C.c_int32(-1), C.c_int32(-1)
    text = C.create_string_buffer(512)
    started = time.monotonic()
    code = lib.rth_test_rank_frames(world, failing_rank, cancel_after_pieces, C.byref(entered), C.byref(traced), text, 512)
    return code, entered.value, traced.value, text.value.decode(), time.monotonic() - started


@pytest.mark.parametrize("world", [1, 2, 8])
def test_every_rank_enters_the_collective_when_no_share_fails(rt, world):
    code, entered, traced, text, _ = run_ranks(rt, world)
    assert (code, entered, traced, text) == (0, world, 8 * world, "")


@pytest.mark.parametrize("world,failing", [(2, 1), (8, 5), (8, 0), (1, 0)])
def test_a_failing_rank_keeps_every_rank_out_of_the_collective(rt, world, failing):
    code, entered, _, text, seconds = run_ranks(rt, world, failing_rank=failing)
    assert code == 0, text                 # 2 would mean: entered by some ranks only, i.e. a hang
    assert entered == 0
    assert f"injected failure of rank {failing}" in text
    if failing > 0:
        assert text.startswith(f"another GPU's share of the frame failed (rank {failing})")
    assert seconds < 1.5                   # nobody waited for the stand-in collective's two-second deadline


def test_giving_the_frame_up_stops_the_other_ranks_photon_ranges(rt):
    code, entered, traced, text, _ = run_ranks(rt, 4, cancel_after_pieces=2)
    assert (code, text) == (0, "")
    assert entered == 4                    # a frame given up is still summed: what was traced is the image
    assert 4 * 2 <= traced <= 4 * 3 + 1    # every rank stopped within a piece of rank 0's second one
    assert traced < 4 * 8
