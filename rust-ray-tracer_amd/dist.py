"""Multi-GPU partition of the render path: one process per GPU, samples-per-pixel slices, one reduce.

Every sample of every pixel is independent (reference ``src/main.rs:91``), so a rank renders a contiguous slice of the
sample indices of all pixels with the same counter-based RNG keying; the only exchange step is the sum of the f32
radiance accumulation buffers on rank 0 (RCCL over xGMI when the tensors are on GPUs, gloo on CPU in the tests),
followed by the tone map there.  Only the root needs the image, hence a reduce, not an all-reduce.

Two partitions: ``split_spp`` cuts a fixed total into ``world`` slices (strong scaling: the frame is the same whatever
the number of GPUs), ``spp_slice`` gives every rank the same count (weak scaling: the frame gets ``world`` times the
samples).  ``rank_slice`` picks between them.
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


def rank_slice(rank: int, world: int, spp: int, scaling: str):
    """(spp_offset, spp of this rank, spp_total of the frame).  scaling "strong": `spp` is the frame's total, split
    over the ranks; "weak": `spp` is what every rank renders."""
    if scaling == "strong":
        if not 0 <= rank < world:
            raise ValueError("bad rank / world")
        off, n = split_spp(spp, world)[rank]
        return off, n, spp
    if scaling == "weak":
        return spp_slice(rank, world, spp)
    raise ValueError(f"unknown scaling {scaling!r}")


def _active():
    import torch.distributed as dist

    # a group of one rank counts: its collectives are trivial, and running them is how the RCCL path is exercised on a
    # box with one GPU
    return dist.is_available() and dist.is_initialized()


def reduce_accum(accum, dst: int = 0, via_host: bool = False):
    """Sums the accumulation tensors of all ranks into rank `dst` (in place there).  No-op in a single process.
    via_host: the tensors live on a GPU but the process group is a CPU one (ranks sharing one GPU in a rehearsal, which
    RCCL refuses): the sum goes through host copies."""
    import torch.distributed as dist

    if not _active():
        return accum
    if via_host:
        h = accum.cpu()
        dist.reduce(h, dst=dst, op=dist.ReduceOp.SUM)
        if dist.get_rank() == dst:
            accum.copy_(h)
    else:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


def all_reduce_value(value: float, op: str, device=None, via_host: bool = False) -> float:
    """max / sum of one float over the ranks (timings, ray counts)."""
    import torch
    import torch.distributed as dist

    if not _active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if via_host or device is None else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return float(t.item())
