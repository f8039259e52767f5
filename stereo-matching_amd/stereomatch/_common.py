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


def empty_tensor(*sizes, dtype: torch.dtype = torch.float32,
                 reuse_tensor: Optional[torch.Tensor] = None,
                 device: torch.device = torch.device("cpu")) -> torch.Tensor:
    """Upstream-named entry point (``_common.py:15-33``): ``reuse_tensor`` when compatible,
    else an uninitialised tensor of the given shape."""
    return output_tensor(sizes, dtype, torch.device(device), reuse=reuse_tensor)


def zeros_tensor_like(base_tensor: torch.Tensor,
                      reuse_tensor: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Upstream-named entry point (``_common.py:36-53``): a zero-filled tensor like
    ``base_tensor``; a compatible ``reuse_tensor`` is zeroed in place and returned."""
    if is_reusable(reuse_tensor, base_tensor.shape, base_tensor.dtype, base_tensor.device):
        return reuse_tensor.zero_()
    return torch.zeros_like(base_tensor)
