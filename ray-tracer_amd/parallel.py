"""Multi-GPU partition of the photon pass.

Photons are independent and every sink is a pure sum (reference DESIGN.md:242-267,
671-676; tests/test_CountEquivalence.cpp:85-135), so a frame of P photons per light is
split by photon INDEX: rank r of N traces indices [r*P/N, (r+1)*P/N) of every light.
Per-photon flux stays Phi / P (the global count, reference src/LightQueue.cpp:16), so
the per-rank camera buffers and counters simply add — one all-reduce(sum) at the end
of the pass, nothing during it.
"""


def shard_range(photons, rank, world):
    """(first, count) of the photon indices rank `rank` of `world` traces."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} of {world}")
    lo = photons * rank // world
    hi = photons * (rank + 1) // world
    return lo, hi - lo


def store_share(capacity, emitter_records, rank, world):
    """Records rank `rank` of `world` may keep of a frame's ONE bounce store of `capacity` records
    (reference src/Renderer.cpp:382-390): the emitter deposits, which go in first and are rank 0's,
    plus an equal part of what they leave.  The parts add up to the capacity (rounded up per rank),
    so a frame whose store overflows keeps as many records on N GPUs as on one."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} of {world}")
    emitters = min(int(emitter_records), int(capacity))
    rest = int(capacity) - emitters
    return max(1, (emitters if rank == 0 else 0) + (rest + world - 1) // world)


def all_reduce_sum(tensors, dist=None):
    """Sum each tensor over all ranks in place (RCCL on GPUs, gloo in CPU tests)."""
    if dist is None:
        import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return tensors
    for t in tensors:
        dist.all_reduce(t)
    return tensors
