"""Sharding of independent stereo pairs over the GPUs of one node.

The reference is single-process, single-GPU ("Parallelism strategies: none").
Stereo pairs are independent, so the only multi-GPU structure the hot path needs is: one
process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI), a contiguous
block of pairs per rank, no collective on the data path, and one all-gather of the int32
disparity maps at the end (8.3 MB per 1080p pair).
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def pair_shard(n_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of the pairs rank ``rank`` of ``world`` owns: contiguous blocks whose
    sizes differ by at most one, earlier ranks taking the larger ones."""
    if world <= 0 or not 0 <= rank < world or n_pairs < 0:
        raise ValueError("bad shard request: n_pairs=%d rank=%d world=%d" % (n_pairs, rank, world))
    base, extra = divmod(n_pairs, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_disparities(local: torch.Tensor, n_pairs: Optional[int] = None,
                       group=None) -> torch.Tensor:
    """All ranks' ``[B_local, H, W]`` int32 disparity maps, concatenated in pair order, on
    every rank.  ``n_pairs`` (total) is needed only when the shards are ragged."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    if world == 1:
        return local
    rank = dist.get_rank(group)
    if local.dim() == 2:
        local = local.unsqueeze(0)
    if n_pairs is None:
        n_pairs = int(local.shape[0]) * world
    sizes = [pair_shard(n_pairs, r, world) for r in range(world)]
    counts = [b - a for a, b in sizes]
    assert counts[rank] == local.shape[0], "local batch does not match pair_shard()"
    local = local.contiguous()
    if len(set(counts)) == 1 and dist.get_backend(group) == "nccl":
        out = torch.empty((world * counts[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                          device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    # Ragged shards (or a backend without all_gather_into_tensor): pad to the largest shard.
    width = max(counts)
    padded = local
    if local.shape[0] < width:
        pad = torch.zeros((width - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                          device=local.device)
        padded = torch.cat([local, pad], dim=0)
    if dist.get_backend(group) != "nccl" and padded.is_cuda:
        # Rehearsal backends (gloo) gather through host memory.
        host = padded.cpu()
        pieces = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(pieces, host, group=group)
        return torch.cat([p[:c] for p, c in zip(pieces, counts)], dim=0).to(padded.device)
    pieces = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(pieces, padded, group=group)
    return torch.cat([p[:c] for p, c in zip(pieces, counts)], dim=0)


def estimate_sharded(pipeline, lefts: torch.Tensor, rights: torch.Tensor, group=None) -> torch.Tensor:
    """Run ``pipeline.estimate`` on this rank's block of the ``[N, H, W]`` pair stack and
    return all N disparity maps (on every rank).  Without an initialised process group this
    is just a batched ``estimate``."""
    n = int(lefts.shape[0])
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    a, b = pair_shard(n, rank, world)
    if b > a:
        local = pipeline.estimate(lefts[a:b], rights[a:b])
        # The gather hands this rank's maps to everybody: the one place where an invalid result
        # (status word of the tiled aggregation: its row sweeps did not have the device's CUs to
        # themselves) must not pass.  Recovery as the C ABI documents it: the paired form.
        # (status() consumes the report, so the re-run below is not refused for it.)
        status = getattr(pipeline, "status", None)
        word = int(status()) if callable(status) else 0
        if word:
            import warnings
            warnings.warn("stereomatch: tiled aggregation reported status %d on rank %d; "
                          "recomputing this shard with the paired form" % (word, rank))
            agg = pipeline.aggregation
            form, agg.form = agg.form, "paired"
            try:
                local = pipeline.estimate(lefts[a:b], rights[a:b])
            finally:
                agg.form = form
    else:
        # Fewer pairs than ranks: this rank owns none.  It still takes part in the gather
        # (with a zero-length block), otherwise the other ranks would wait for it for ever.
        local = torch.empty((0,) + tuple(lefts.shape[1:]), dtype=torch.int32, device=lefts.device)
    return gather_disparities(local, n_pairs=n, group=group)
