"""Output-buffer policy shared by the op wrappers.

Mirrors the behaviour of upstream ``stereomatch/_common.py:9-53``: an output tensor
handed in by the caller is reused (and returned, same object) when its shape, dtype
and device all match what the op produces; otherwise a fresh tensor is allocated.
``Pipeline`` relies on that identity to keep its three buffers alive across frames
(upstream ``stereomatch/pipeline.py:80-92``).
"""
from typing import Optional, Sequence

import torch


def is_reusable(candidate: Optional[torch.Tensor], shape: Sequence[int],
                dtype: torch.dtype, device: torch.device) -> bool:
    return (candidate is not None
            and tuple(candidate.shape) == tuple(shape)
            and candidate.dtype == dtype
            and candidate.device == device
            and candidate.is_contiguous())


def output_tensor(shape: Sequence[int], dtype: torch.dtype, device: torch.device,
                  reuse: Optional[torch.Tensor] = None, zero_fresh: bool = False) -> torch.Tensor:
    """``reuse`` itself when compatible, else a new tensor (zero-filled on request)."""
    if is_reusable(reuse, shape, dtype, device):
        return reuse
    if zero_fresh:
        return torch.zeros(tuple(shape), dtype=dtype, device=device)
    return torch.empty(tuple(shape), dtype=dtype, device=device)
