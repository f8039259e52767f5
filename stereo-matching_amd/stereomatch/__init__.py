"""MI355X-native stereo cost-volume pipeline with the ``stereomatch`` Python API.

Same callables as otaviog/stereo-matching (``stereomatch/__init__.py:4`` imports
``aggregation``, ``cost``, ``disparity_reduce``): ``cost.SSD``, ``cost.Birchfield``,
``aggregation.Semiglobal``, ``disparity_reduce.WinnerTakesAll``,
``disparity_reduce.DynamicProgramming`` and ``pipeline.Pipeline``.  Tensors are
PyTorch-ROCm device tensors; the arithmetic runs in ``libstereomatch_hip.so``
(hand-written HIP for gfx950, C ABI in ``include/stereomatch_hip.h``).  There is no
CPU fallback: CPU tensors raise ``RuntimeError``.
"""
from . import (aggregation, cost, cuda_texture, disparity_reduce, distributed, numeric,  # noqa: F401
               pipeline)
from .pipeline import Pipeline  # noqa: F401

__all__ = ["aggregation", "cost", "disparity_reduce", "distributed", "numeric", "pipeline",
           "Pipeline"]
