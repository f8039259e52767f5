"""Name -> method registry and ``create_pipeline`` (counterpart of upstream
``stereomatch/cli_common.py:9-52``; the CUDA-texture cost has no CDNA4 equivalent and is
not registered)."""
from .aggregation import Semiglobal
from .cost import SSD, Birchfield
from .disparity_reduce import DynamicProgramming, WinnerTakesAll
from .pipeline import Pipeline

COST_METHODS = {"ssd": SSD, "birchfield": Birchfield}
AGGREGATION_METHODS = {"sgm": Semiglobal}
DISPARITY_METHODS = {"wta": WinnerTakesAll, "dyn": DynamicProgramming}


def create_pipeline(cost_method: str, disp_method: str, aggr_method: str = None,
                    max_disparity: int = 32) -> Pipeline:
    """``Pipeline`` from method names: cost in ``COST_METHODS``, reducer in
    ``DISPARITY_METHODS``, optional aggregation in ``AGGREGATION_METHODS`` (any other value,
    e.g. ``None``, means no aggregation -- as upstream)."""
    aggregation = AGGREGATION_METHODS.get(aggr_method)
    return Pipeline(COST_METHODS[cost_method](max_disparity), DISPARITY_METHODS[disp_method](),
                    aggregation=aggregation() if aggregation is not None else None)
