"""Matching-cost volumes (counterpart of upstream ``stereomatch/cost.py``).

``SSD`` -> ``stm_ssd_f32`` (reference ``CostOps.compute_ssd``, src/ssd.cu:64-81)
``Birchfield`` -> ``stm_birchfield_f32`` (``CostOps.compute_birchfield``,
src/birchfield_cost.cu:153-181).  Images are ``[H, W]`` (the reference's shape) or a
batch ``[B, H, W]``; volumes ``[H, W, D]`` / ``[B, H, W, D]`` float32.
"""
from typing import Optional

import torch

from . import _native
from ._common import output_tensor


class SSD:
    """Sum of squared differences over a ``2k x 2k`` window (``k = kernel_size``).

    Same constructor as upstream ``cost.py:21-25``.  Only float32 images and a float32
    volume are accepted, which is all the reference's dispatch ever lets through
    (anything else raises "SSDKernel: type not supported").

    ``mode``: ``"auto"`` (default) lets a device-side classifier choose between the
    ordered 4k^2-tap sum and separable box sums -- the result is bit-identical to the
    reference either way; ``"ordered"`` / ``"box"`` force one path (``"box"`` is exact
    only for integer-valued images, see ``include/stereomatch_hip.h``).
    """

    def __init__(self, max_disparity: int, kernel_size: int = 7,
                 cost_volume_dtype: torch.dtype = torch.float, mode: str = "auto"):
        if mode not in _native.STM_SSD_MODES:
            raise ValueError("SSD mode must be one of %s" % sorted(_native.STM_SSD_MODES))
        self.max_disparity = max_disparity
        self.kernel_size = kernel_size
        self.cost_volume_dtype = cost_volume_dtype
        self.mode = mode
        self._workspace = None

    def __call__(self, left_image: torch.Tensor, right_image: torch.Tensor,
                 cost_volume: Optional[torch.Tensor] = None) -> torch.Tensor:
        _native.require_gpu(left_image, right_image)
        if (left_image.dtype != torch.float32 or right_image.dtype != torch.float32
                or self.cost_volume_dtype != torch.float32):
            raise RuntimeError("SSDKernel: type not supported")
        if left_image.shape != right_image.shape:
            raise RuntimeError("stereomatch: left and right image shapes differ")
        batch, (height, width) = _native.split_batch(left_image, 2)
        shape = tuple(left_image.shape) + (self.max_disparity,)
        device = left_image.device
        cost_volume = output_tensor(shape, torch.float32, device, reuse=cost_volume)
        left = left_image.contiguous()
        right = right_image.contiguous()
        lib = _native.lib()
        ws_bytes = lib.stm_ssd_workspace_bytes(height, width, self.max_disparity,
                                               self.kernel_size, batch)
        if (self._workspace is None or self._workspace.device != device
                or self._workspace.numel() < ws_bytes):
            self._workspace = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            _native.check(lib.stm_ssd_f32(
                left.data_ptr(), right.data_ptr(), cost_volume.data_ptr(), height, width,
                self.max_disparity, self.kernel_size, _native.STM_SSD_MODES[self.mode],
                self._workspace.data_ptr(), self._workspace.numel(), batch,
                _native.stream_ptr(device)))
        return cost_volume


class SSDTexture:
    """Upstream ``cost.py:51-77`` samples the images through CUDA texture objects.
    CDNA4 has no texture units (hipcc rejects ``tex2D`` for gfx950), so this variant is
    not provided; ``SSD`` computes the same volume."""

    def __init__(self, *args, **kwargs):
        raise RuntimeError("SSDTexture needs texture hardware, which MI355X (gfx950) does not "
                           "have; use stereomatch.cost.SSD")


class Birchfield:
    """Birchfield-Tomasi cost with the reference's fixed 8-tap window.

    ``kernel_size`` is accepted and ignored, as upstream (``cost.py:88-99`` stores it
    under a misspelt attribute and passes the constant 4).  Entries with ``d > x`` are
    never written (reference behaviour); a freshly allocated volume is zero-filled so
    they read as 0, a reused one keeps its previous contents there.
    """

    def __init__(self, max_disparity: int, kernel_size: int = 7, indexing: str = "reference"):
        if indexing not in _native.STM_BT_INDEXING:
            raise ValueError("Birchfield indexing must be one of %s" % sorted(_native.STM_BT_INDEXING))
        self.max_disparity = max_disparity
        self.kernel_size = kernel_size
        # "reference": the CPU functor's shifted flat indexing (the parity target);
        # "padded": the CUDA functor's (sample p with zero row borders) -- an opt-in correction.
        self.indexing = indexing

    def __call__(self, left_image: torch.Tensor, right_image: torch.Tensor,
                 cost_volume: Optional[torch.Tensor] = None) -> torch.Tensor:
        _native.require_gpu(left_image, right_image)
        if left_image.shape != right_image.shape:
            raise RuntimeError("stereomatch: left and right image shapes differ")
        batch, (height, width) = _native.split_batch(left_image, 2)
        shape = tuple(left_image.shape) + (self.max_disparity,)
        device = left_image.device
        cost_volume = output_tensor(shape, torch.float32, device, reuse=cost_volume,
                                    zero_fresh=True)
        left = left_image.float().contiguous()
        right = right_image.float().contiguous()
        with torch.cuda.device(device):
            _native.check(_native.lib().stm_birchfield_ex_f32(
                left.data_ptr(), right.data_ptr(), cost_volume.data_ptr(), height, width,
                self.max_disparity, _native.STM_BT_INDEXING[self.indexing], batch,
                _native.stream_ptr(device)))
        return cost_volume
