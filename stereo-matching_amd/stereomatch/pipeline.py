"""Cost -> (aggregation) -> reduce composition (counterpart of upstream
``stereomatch/pipeline.py:36-94``)."""
from typing import Callable, Optional

import torch

from .aggregation import Semiglobal
from .disparity_reduce import WinnerTakesAll
from .numeric import is_power_of_two

CostFunction = Callable[..., torch.Tensor]
AggregationFunction = Callable[[torch.Tensor, torch.Tensor, Optional[torch.Tensor]], torch.Tensor]
DisparityReduceFunction = Callable[[torch.Tensor, Optional[torch.Tensor]], torch.Tensor]


class Pipeline:
    """``Pipeline(cost, disparity_reduce, aggregation=None).estimate(left, right)``.

    Accepts any three callables with the op wrappers' signatures.  The cost volume, the
    aggregated volume and the disparity map of the previous call are offered back to the
    ops as output buffers, so a steady stream of equally sized frames allocates nothing
    (upstream ``pipeline.py:65-67,80-92``).
    """

    def __init__(self, cost: CostFunction, disparity_reduce: DisparityReduceFunction,
                 aggregation: Optional[AggregationFunction] = None, fuse: bool = True):
        # fuse: when the aggregation is this package's Semiglobal (paired form) and the
        # reducer its WinnerTakesAll, run them as one fused call: same disparity map, but the
        # aggregated volume is never written out, so `_aggregation_volume` is scratch then.
        self.fuse = fuse
        self.cost = cost
        self.disparity_reduce = disparity_reduce
        self.aggregation = aggregation
        self._cost_volume = None
        self._aggregation_volume = None
        self._disparity_image = None

    def _can_fuse(self) -> bool:
        return (self.fuse and type(self.aggregation) is Semiglobal
                and not self.aggregation.single_passes
                and type(self.disparity_reduce) is WinnerTakesAll
                and self._cost_volume.is_cuda and self._cost_volume.dtype == torch.float32)

    def status(self) -> int:
        """Status word of the last ``estimate`` (waits for it; 0 = ok).  Only this package's
        tiled aggregation can report anything else: a set bit means that call's disparity map
        is invalid and should be recomputed with ``Semiglobal(form="paired")`` (aggregation.py,
        "status word").  ``estimate`` itself never blocks; it raises for a failed previous
        call as soon as that call's word has reached the host."""
        status = getattr(self.aggregation, "status", None)
        return int(status()) if callable(status) else 0

    def estimate(self, left_image: torch.Tensor, right_image: torch.Tensor) -> torch.Tensor:
        self._cost_volume = self.cost(left_image, right_image, cost_volume=self._cost_volume)
        if self.aggregation is None:
            volume = self._cost_volume
        elif self._can_fuse():
            if (self.disparity_reduce.require_power_of_two
                    and not is_power_of_two(int(self._cost_volume.shape[-1]))):
                raise RuntimeError("Winner takes all requires max disparity "
                                   "(`cost_volume.size(-1)`) to be a power of 2.")
            self._aggregation_volume, self._disparity_image = self.aggregation.aggregate_and_reduce(
                self._cost_volume, left_image.to(self._cost_volume.device),
                self._aggregation_volume, self._disparity_image)
            return self._disparity_image
        else:
            # The image follows the volume's device (upstream pipeline.py:85-87).
            self._aggregation_volume = self.aggregation(
                self._cost_volume, left_image.to(self._cost_volume.device),
                self._aggregation_volume)
            volume = self._aggregation_volume
        self._disparity_image = self.disparity_reduce(volume, self._disparity_image)
        return self._disparity_image
