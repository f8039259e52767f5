"""Cost -> (aggregation) -> reduce composition (counterpart of upstream
``stereomatch/pipeline.py:36-94``)."""
from typing import Callable, Optional

import torch

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
                 aggregation: Optional[AggregationFunction] = None):
        self.cost = cost
        self.disparity_reduce = disparity_reduce
        self.aggregation = aggregation
        self._cost_volume = None
        self._aggregation_volume = None
        self._disparity_image = None

    def estimate(self, left_image: torch.Tensor, right_image: torch.Tensor) -> torch.Tensor:
        self._cost_volume = self.cost(left_image, right_image, cost_volume=self._cost_volume)
        if self.aggregation is None:
            volume = self._cost_volume
        else:
            # The image follows the volume's device (upstream pipeline.py:85-87).
            self._aggregation_volume = self.aggregation(
                self._cost_volume, left_image.to(self._cost_volume.device),
                self._aggregation_volume)
            volume = self._aggregation_volume
        self._disparity_image = self.disparity_reduce(volume, self._disparity_image)
        return self._disparity_image
