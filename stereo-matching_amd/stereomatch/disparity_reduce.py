"""Cost volume -> disparity map reducers (counterpart of upstream
``stereomatch/disparity_reduce.py``)."""
from typing import Optional

import torch

from . import _native
from ._common import output_tensor
from .numeric import is_power_of_two


class WinnerTakesAll:
    """Disparity with the lowest cost; the lowest index wins ties (the reference CPU
    rule, src/winners_take_all.cu:21-40).

    Like upstream (``disparity_reduce.py:40-43``, tested by
    ``tests/test_winners_take_all.py:42-48``) a device tensor whose disparity count is
    not a power of two raises ``RuntimeError``.  The kernel itself has no such limit;
    pass ``require_power_of_two=False`` to lift the check.
    """

    def __init__(self, require_power_of_two: bool = True):
        self.require_power_of_two = require_power_of_two

    def __call__(self, cost_volume: torch.Tensor,
                 disparity_img: Optional[torch.Tensor] = None) -> torch.Tensor:
        _native.require_gpu(cost_volume)
        batch, (height, width, max_disparity) = _native.split_batch(cost_volume, 3)
        device = cost_volume.device
        disparity_img = output_tensor(cost_volume.shape[:-1], torch.int32, device,
                                      reuse=disparity_img)
        if self.require_power_of_two and not is_power_of_two(max_disparity):
            raise RuntimeError("Winner takes all requires max disparity "
                               "(`cost_volume.size(-1)`) to be a power of 2.")
        _native.require_float32(cost_volume, "RunWinnersTakeAll", "src/winners_take_all.cu:99,108")
        volume = cost_volume.contiguous()
        with torch.cuda.device(device):
            _native.check(_native.lib().stm_wta_f32(
                volume.data_ptr(), disparity_img.data_ptr(), height, width, max_disparity,
                batch, _native.stream_ptr(device)))
        return disparity_img


class DynamicProgramming:
    """Scanline dynamic programming (reference ``RunDynamicProgramming``,
    src/dynamic_programming.cu:184-225).

    ``mode="reference"`` reproduces the reference CPU path including its transposed
    initial read (needs ``D <= H <= W``); ``mode="corrected"`` starts every row from its
    own first column.  The int8 path volume and the per-row final costs are scratch
    allocated per call, as upstream (``disparity_reduce.py:77-84``); the last ones are
    kept on the instance as ``path_volume`` / ``row_final_costs`` for inspection.
    """

    def __init__(self, mode: str = "reference"):
        if mode not in _native.STM_DP_MODES:
            raise ValueError("DynamicProgramming mode must be one of %s"
                             % sorted(_native.STM_DP_MODES))
        self.mode = mode
        self.path_volume = None
        self.row_final_costs = None

    def __call__(self, cost_volume: torch.Tensor,
                 disparity_img: Optional[torch.Tensor] = None) -> torch.Tensor:
        _native.require_gpu(cost_volume)
        batch, (height, width, max_disparity) = _native.split_batch(cost_volume, 3)
        _native.require_float32(cost_volume, "RunDynamicProgramming",
                                "src/dynamic_programming.cu:199,213")
        device = cost_volume.device
        disparity_img = output_tensor(cost_volume.shape[:-1], torch.int32, device,
                                      reuse=disparity_img)
        self.path_volume = output_tensor(cost_volume.shape, torch.int8, device,
                                         reuse=self.path_volume)
        self.row_final_costs = output_tensor(tuple(cost_volume.shape[:-2]) + (max_disparity,),
                                             torch.float32, device, reuse=self.row_final_costs)
        volume = cost_volume.contiguous()
        with torch.cuda.device(device):
            _native.check(_native.lib().stm_dp_f32(
                volume.data_ptr(), self.path_volume.data_ptr(), self.row_final_costs.data_ptr(),
                disparity_img.data_ptr(), height, width, max_disparity,
                _native.STM_DP_MODES[self.mode], batch, _native.stream_ptr(device)))
        return disparity_img
