#!/usr/bin/env python3
"""Times the aggregation's two forms (paired / tiled) on one resident batch, fused with the
winner-takes-all and unfused, and checks that they agree bit for bit.

    python tools/sgm_forms.py 1080 1920 256 8     # H W D batch
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stereo-matching_amd")):
    sys.path.insert(0, p)
import stereomatch  # noqa: E402


def main():
    H, W, D, B = [int(a) for a in sys.argv[1:5]]
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    base = rng.integers(0, 256, size=(B, H, W + 64)).astype(np.float32)
    L = torch.from_numpy(np.ascontiguousarray(base[:, :, 32:32 + W])).to(dev)
    R = torch.from_numpy(np.ascontiguousarray(base[:, :, 39:39 + W])).to(dev)
    C = stereomatch.cost.SSD(D)(L, R)
    res = {}
    forms = os.environ.get("FORMS", "paired,tiled,walks,skew").split(",")
    forms = [f for f in forms if f != "skew" or D in (128, 256)]
    for form in forms:
        sgm = stereomatch.aggregation.Semiglobal(form=form)
        for fused in (True, False):
            vol = disp = None
            ms = []
            for _ in range(reps + 1):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                if fused:
                    vol, disp = sgm.aggregate_and_reduce(C, L, vol, disp)
                else:
                    vol = sgm(C, L, vol)
                b.record()
                torch.cuda.synchronize()
                ms.append(a.elapsed_time(b))
            res[(form, fused)] = (disp.clone() if fused else vol.clone())
            n = H * W * D * B
            t = float(np.median(ms[1:]))
            print("%-6s %-7s %8.3f ms  (%.3f ms/pair, %.1f GB/s algorithmic = %.3f of 8 TB/s) status %d"
                  % (form, "fused" if fused else "unfused", t, t / B, 8.0 * n / t / 1e6,
                     8.0 * n / t / 1e6 / 8000.0, sgm.status()), flush=True)
        del sgm
        torch.cuda.empty_cache()
    for form in [f for f in forms if f != "paired" and "paired" in forms]:
        print(form, "fused disparities equal:", bool(torch.equal(res[("paired", True)], res[(form, True)])),
              " unfused volumes equal:", bool(torch.equal(res[("paired", False)].view(torch.int32),
                                                          res[(form, False)].view(torch.int32))))


if __name__ == "__main__":
    main()
