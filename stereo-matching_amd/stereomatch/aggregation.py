"""Cost-volume aggregation (counterpart of upstream ``stereomatch/aggregation.py``)."""
from typing import Optional

import torch

from . import _native
from ._common import output_tensor


class Semiglobal:
    """Semiglobal matching aggregation with the gradient-adaptive second penalty.

    ``Semiglobal(penalty1=0.1, penalty2=0.2)`` as upstream ``aggregation.py:26-33``;
    calls ``stm_sgm_f32`` (reference ``AggregationOps.run_semiglobal``,
    src/semiglobal.cpp:167-197).  The result equals the reference CPU path bit for bit:
    eight path directions summed in its path-table order, including its quirks for the
    inverse diagonals and the minimum over ``d < D-1``.

    Non-finite inputs: that holds for every volume whose pixels all have a finite cost at
    some ``d < D-1`` (all volumes ``SSD`` and ``Birchfield`` produce).  A pixel that is +inf
    at every ``d < D-1`` makes the recurrence compute ``(C + inf) - inf``; NaN and -inf then
    spread along every path through it, and with ``exact_nonfinite=False`` (default) which of
    them lands on which element is unspecified.  ``exact_nonfinite=True``
    (``STM_SGM_EXACT_NONFINITE``) reproduces the reference's placement too, 1-3 % slower.

    The reference zero-fills the output and accumulates into it
    (``aggregation.py:51``); here the first pass overwrites the output instead, which
    gives the same values without the extra fill and read.
    """

    def __init__(self, penalty1: float = 0.1, penalty2: float = 0.2, single_passes: bool = False,
                 paths: str = "reference", full_min: bool = False,
                 exact_nonfinite: bool = False, form: str = "auto", cooperative: bool = False):
        if paths not in ("reference", "hirschmuller"):
            raise ValueError("Semiglobal paths must be 'reference' or 'hirschmuller'")
        self.penalty1 = penalty1
        self.penalty2 = penalty2
        # Corrected-algorithm options (extensions; the defaults reproduce the reference):
        # paths="hirschmuller": every pixel receives each of the eight directions exactly once
        # (the reference's inverse diagonals skip 17-44 % of the pixels);
        # full_min=True: the per-step minimum covers all D disparities (Q5).
        self.paths = paths
        self.full_min = full_min
        self.exact_nonfinite = exact_nonfinite
        # "auto": the library picks per call between its paired form (a wave per path) and its
        # tiled form (8x8 tiles rebuilt from border states, D = 64/128/256), the latter fed by row
        # sweeps (large batches) or by per-path walks (single pairs, small batches);
        # "paired" / "tiled" / "walks" force one (where it covers the call).  Same bits.
        if form not in _native.STM_SGM_FORMS:
            raise ValueError("Semiglobal form must be one of %s" % sorted(_native.STM_SGM_FORMS))
        self.form = form
        # cooperative=True (STM_SGM_COOPERATIVE): the tiled form's row sweeps are launched as
        # cooperative kernels, which the runtime runs one at a time per device -- for processes
        # that run tiled aggregations on several streams of one device.  ~60 us per sweep launch
        # and the call stops being asynchronous; by default the device is expected to be this
        # call's while it runs (include/stereomatch_hip.h, "Residency contract").
        self.cooperative = cooperative
        # False (default): a direction and its inverse share one pass over the sum (less HBM
        # traffic, larger workspace); True: eight separate passes.  Same bits either way.
        self.single_passes = single_passes
        self._workspace = None
        self._last_tiled = False   # the last call took a tiled form (behind sweeps or walks)
        self._last_form = _native.STM_SGM_FORM_PAIRED
        # Status word of the last tiled call, on its way to the host (see status()).
        self._status_host = None
        self._status_event = None
        self._status_word = None   # the last call's word once status() has read (consumed) it
        # False: the next call does not raise for a failed previous one by itself -- for callers
        # inside a collective context, which read status() on every rank at a common point and
        # agree on the verdict (distributed.estimate_sharded, bench.py): a raise on one rank
        # alone would leave the others waiting in the next collective.
        self.raise_deferred = True

    def __call__(self, cost_volume: torch.Tensor, left_image: torch.Tensor,
                 sga_volume: Optional[torch.Tensor] = None) -> torch.Tensor:
        sga_volume, _ = self._run(cost_volume, left_image, sga_volume, None, fused=False)
        return sga_volume

    def aggregate_and_reduce(self, cost_volume: torch.Tensor, left_image: torch.Tensor,
                             sga_volume: Optional[torch.Tensor] = None,
                             disparity_img: Optional[torch.Tensor] = None,
                             keep_sum: bool = False):
        """Aggregation and winner-takes-all in one call (``stm_sgm_wta_f32``): the last
        direction pair reduces each finished row while it is in registers.  Returns
        ``(volume, disparity)``; the disparity equals ``WinnerTakesAll()(self(cost, left))``
        bit for bit.  Unless ``keep_sum`` the returned volume is scratch (it holds the sum of
        the first six directions)."""
        if self.single_passes:
            raise RuntimeError("Semiglobal.aggregate_and_reduce needs the paired form "
                               "(single_passes=False)")
        return self._run(cost_volume, left_image, sga_volume, disparity_img, fused=True,
                         keep_sum=keep_sum)

    def _run(self, cost_volume, left_image, sga_volume, disparity_img, fused, keep_sum=False):
        _native.require_gpu(cost_volume, left_image)
        _native.require_float32(cost_volume, "RunSemiglobal", "src/semiglobal.cpp:182")
        batch, (height, width, max_disparity) = _native.split_batch(cost_volume, 3)
        if tuple(left_image.shape) != tuple(cost_volume.shape[:-1]):
            raise RuntimeError("stereomatch: left image shape %s does not match cost volume %s"
                               % (tuple(left_image.shape), tuple(cost_volume.shape)))
        device = cost_volume.device
        if sga_volume is not None and sga_volume.data_ptr() == cost_volume.data_ptr():
            sga_volume = None  # in-place aggregation is not possible
        sga_volume = output_tensor(cost_volume.shape, torch.float32, device, reuse=sga_volume)
        cost = cost_volume.contiguous()
        image = left_image.float().contiguous()
        lib = _native.lib()
        flags = _native.STM_SGM_SINGLE_PASSES if self.single_passes else 0
        if self.paths == "hirschmuller":
            flags |= _native.STM_SGM_TRUE_PATHS
        if self.full_min:
            flags |= _native.STM_SGM_FULL_MIN
        if self.exact_nonfinite:
            flags |= _native.STM_SGM_EXACT_NONFINITE
        flags |= _native.STM_SGM_FORMS[self.form]
        if self.cooperative:
            flags |= _native.STM_SGM_COOPERATIVE
        if fused:
            disparity_img = output_tensor(cost_volume.shape[:-1], torch.int32, device,
                                          reuse=disparity_img)
            if keep_sum:
                flags |= _native.STM_SGM_KEEP_SUM
        # A previous call whose row sweeps ran into their bounded waits produced an invalid
        # result: say so as soon as its status word has arrived (never blocks).
        self._raise_if_failed(block=False)
        with torch.cuda.device(device):
            # (form selection and workspace layout depend on the current device's CU count)
            ws_bytes = lib.stm_sgm_workspace_bytes(height, width, max_disparity, batch, flags)
            self._last_form = int(lib.stm_sgm_form(height, width, max_disparity, batch, flags))
            self._last_tiled = self._last_form != _native.STM_SGM_FORM_PAIRED
            if ws_bytes and (self._workspace is None or self._workspace.device != device
                             or self._workspace.numel() < ws_bytes):
                self._workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
            ws_ptr = self._workspace.data_ptr() if (ws_bytes and self._workspace is not None) else None
            if fused:
                _native.check(lib.stm_sgm_wta_f32(
                    cost.data_ptr(), image.data_ptr(), sga_volume.data_ptr(),
                    disparity_img.data_ptr(), height, width, max_disparity, float(self.penalty1),
                    float(self.penalty2), flags, ws_ptr, ws_bytes, batch,
                    _native.stream_ptr(device)))
            else:
                _native.check(lib.stm_sgm_f32(
                    cost.data_ptr(), image.data_ptr(), sga_volume.data_ptr(), height, width,
                    max_disparity, float(self.penalty1), float(self.penalty2), flags, ws_ptr,
                    ws_bytes, batch, _native.stream_ptr(device)))
            self._post_status(device)
        return sga_volume, disparity_img

    # ---- status word of the tiled form (include/stereomatch_hip.h, "Workspace header") -------
    # The row sweeps of the tiled form exchange values between workgroups that must all be
    # resident; their waits are bounded, and a wait that expires (the device's CUs were held by
    # something else for seconds) sets bit 0 of the workspace's first word: that call's results
    # are invalid.  The entry points never synchronise, so the word is brought to the host
    # asynchronously behind every tiled call and looked at (a) without blocking at the start of
    # the next call, (b) by status() / check(), which wait for it -- the callers' existing
    # synchronisation points (Pipeline.status(), distributed.estimate_sharded, bench.py).
    def _post_status(self, device) -> None:
        self._status_event = None
        self._status_word = None
        if (self._last_form not in (_native.STM_SGM_FORM_SWEEPS, _native.STM_SGM_FORM_SKEW) or self._workspace is None
                or self._workspace.numel() < 4):
            return  # only the row sweeps have waits that can expire
        if torch.cuda.is_current_stream_capturing():
            return  # (a captured call is checked through status(), which reads the device word)
        if self._status_host is None:
            self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._status_host.copy_(self._workspace[:4].view(torch.int32), non_blocking=True)
        self._status_event = torch.cuda.Event()
        self._status_event.record(torch.cuda.current_stream(device))

    def _raise_if_failed(self, block: bool) -> None:
        ev = self._status_event
        if ev is None or not (block or ev.query()) or not (block or self.raise_deferred):
            return
        ev.synchronize()
        self._status_event = None
        word = int(self._status_host.item())
        if word:
            raise RuntimeError(
                "Semiglobal: the previous tiled aggregation's results are invalid (status word %d: "
                "its row sweeps waited ~2 s for workgroups that were not resident -- the device's "
                "CUs were held by other work); re-run that call with Semiglobal(form='paired')" % word)

    def status(self) -> int:
        """Status word of the last call (waits for that call): 0 = ok; bit 0 = a bounded wait
        inside the tiled form's row sweeps expired and that call's results are invalid.  The
        paired form and the walk-fed tiles have no waits and always report 0.

        Reading the word CONSUMES the pending report: the caller has been told, so the next
        call on this op does not raise for it again (the documented recovery -- re-running the
        call with ``form="paired"`` -- is such a next call)."""
        if (self._last_form not in (_native.STM_SGM_FORM_SWEEPS, _native.STM_SGM_FORM_SKEW) or self._workspace is None
                or self._workspace.numel() < 4):
            return 0
        if self._status_event is not None:
            self._status_event.synchronize()
            self._status_event = None
            self._status_word = int(self._status_host.item())
        if self._status_word is None:  # (a captured call: nothing was posted)
            torch.cuda.synchronize(self._workspace.device)
            self._status_word = int(self._workspace[:4].view(torch.int32).item())
        return self._status_word

    def check(self) -> None:
        """Waits for the last call and raises RuntimeError if its status word is set."""
        word = self.status()
        if word:
            raise RuntimeError("Semiglobal: the last tiled aggregation's results are invalid "
                               "(status word %d); re-run with Semiglobal(form='paired')" % word)
