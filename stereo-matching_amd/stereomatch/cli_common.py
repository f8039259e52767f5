"""Build a ``Pipeline`` from method names.

Same vocabulary as the reference's command-line tools (``ssd`` / ``birchfield`` costs, ``sgm``
aggregation, ``wta`` / ``dyn`` reducers; upstream ``stereomatch/cli_common.py:9-52``).  The
CUDA-texture cost has no CDNA4 equivalent and is not offered.  Extra keyword arguments are
routed to the constructors that accept them, so the MI355X-specific options are reachable from
the same entry point, e.g. ``create_pipeline("ssd", "wta", "sgm", 128, paths="hirschmuller")``.
"""
import inspect
from typing import Dict, Optional

from . import aggregation as _aggregation
from . import cost as _cost
from . import disparity_reduce as _reduce
from .pipeline import Pipeline

_REGISTRY: Dict[str, Dict[str, type]] = {
    "cost": {"ssd": _cost.SSD, "birchfield": _cost.Birchfield},
    "aggregation": {"sgm": _aggregation.Semiglobal},
    "disparity": {"wta": _reduce.WinnerTakesAll, "dyn": _reduce.DynamicProgramming},
}
# Upstream-compatible views of the registry.
COST_METHODS = _REGISTRY["cost"]
AGGREGATION_METHODS = _REGISTRY["aggregation"]
DISPARITY_METHODS = _REGISTRY["disparity"]


def available_methods() -> Dict[str, list]:
    """Names accepted by :func:`create_pipeline`, per stage."""
    return {stage: sorted(table) for stage, table in _REGISTRY.items()}


def _construct(cls, *positional, **options):
    accepted = inspect.signature(cls.__init__).parameters
    return cls(*positional, **{k: v for k, v in options.items() if k in accepted})


def create_pipeline(cost_method: str, disp_method: str, aggr_method: Optional[str] = None,
                    max_disparity: int = 32, **options) -> Pipeline:
    """``cost_method`` and ``disp_method`` must be registered names (``KeyError`` otherwise);
    an unregistered ``aggr_method`` -- typically ``None`` -- means "no aggregation", which is
    how the reference treats it.  ``max_disparity`` defaults to 32 as upstream."""
    stage_cost = _construct(COST_METHODS[cost_method], max_disparity, **options)
    stage_reduce = _construct(DISPARITY_METHODS[disp_method], **options)
    aggregation_cls = AGGREGATION_METHODS.get(aggr_method)
    stage_aggregation = _construct(aggregation_cls, **options) if aggregation_cls else None
    return Pipeline(stage_cost, stage_reduce, aggregation=stage_aggregation)
