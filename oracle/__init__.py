"""CPU oracle for the stereo cost-volume hot path -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``.  The product package (``stereo-matching_amd/stereomatch``)
never imports it.  See ``stm_oracle.c`` for the reference citations and
``tests/test_oracle_pinned.py`` for how it is pinned.
"""
from .oracle import (  # noqa: F401
    build, lib, ssd, ssd_entries, birchfield, sgm, sgm_paths, wta, dp,
    synthetic_pair, digest16,
)
