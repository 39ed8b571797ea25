"""Multi-GPU partition of the render path: one process per GPU, samples-per-pixel slices, one reduce.

Every sample of every pixel is independent (reference ``src/main.rs:91``), so rank ``g`` of ``G`` renders the sample
indices ``[g * spp, (g + 1) * spp)`` of all pixels with the same counter-based RNG keying; the only exchange step is
the sum of the f32 radiance accumulation buffers on rank 0 (RCCL over xGMI when the tensors are on GPUs, gloo on CPU
in the tests), followed by the tone map there.  Only the root needs the image, hence a reduce, not an all-reduce.
"""
from __future__ import annotations


def spp_slice(rank: int, world: int, spp_per_gpu: int):
    """(spp_offset, spp, spp_total) of a rank under weak scaling: per-GPU work fixed, the image gets world * spp samples."""
    if not (0 <= rank < world) or spp_per_gpu <= 0:
        raise ValueError("bad rank / world / spp")
    return rank * spp_per_gpu, spp_per_gpu, world * spp_per_gpu


def split_spp(total_spp: int, world: int):
    """Strong-scaling split of a fixed total: list of (spp_offset, spp) per rank, remainders to the low ranks."""
    if world <= 0 or total_spp < 0:
        raise ValueError("bad world / spp")
    base, rem = divmod(total_spp, world)
    out, off = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((off, n))
        off += n
    return out


def reduce_accum(accum, dst: int = 0):
    """Sums the accumulation tensors of all ranks into rank `dst` (in place there).  No-op in a single process."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum
