#!/usr/bin/env python3
"""Accuracy of the reference-exact aggregation vs the corrected options on synthetic
scenes with known ground truth (in the spirit of upstream workflows/evaluation/workflow.py:
RMSE / average error / bad-pixel rates).  Run on a GPU box:  python tools/accuracy_report.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-matching_amd"))
import numpy as np
import torch
import stereomatch
from stereomatch.data import disparity_metrics


def scene(H, W, seed, noise=10.0):
    """Textured background at disparity 6 with two fronto-parallel boxes at 18 and 30; the
    right image is the left one warped by the ground truth (occluded pixels keep texture)."""
    rng = np.random.default_rng(seed)
    gt = np.full((H, W), 6, dtype=np.int32)
    gt[H // 5: H // 2, W // 6: W // 2] = 18
    gt[H // 2: 4 * H // 5, W // 2: 5 * W // 6] = 30
    tex = rng.integers(0, 256, size=(H + 8, W + 72)).astype(np.float32)
    # Low-pass the texture (9x9 box) so that neighbouring grey values are close and the
    # gradient-adaptive penalty does not collapse to P1 everywhere.
    acc = np.zeros((H, W + 64), dtype=np.float32)
    for dy in range(9):
        for dx in range(9):
            acc += tex[dy:dy + H, dx:dx + W + 64]
    tex = np.round((acc / 81.0 - 128.0) * 6.0 + 128.0).clip(0, 255)
    left = np.zeros((H, W), dtype=np.float32)
    right = rng.integers(0, 256, size=(H, W)).astype(np.float32)
    for d in (6, 18, 30):                      # paint far to near
        ys, xs = np.nonzero(gt == d)
        left[ys, xs] = tex[ys, xs + d]
        xr = xs - d
        ok = xr >= 0
        right[ys[ok], xr[ok]] = left[ys[ok], xs[ok]]
    right = np.clip(right + rng.normal(0, noise, right.shape), 0, 255).round().astype(np.float32)
    return left, right, gt.astype(np.float32)


def main():
    dev = torch.device("cuda:0")
    rows = []
    for (H, W) in ((375, 450), (540, 960)):
        for seed in range(3):
            L, R, gt = scene(H, W, seed)
            # Grey values scaled to [0, 1] (upstream's tests/conftest.py:24-29 does the same), which
            # puts the SSD costs on the scale of the default penalties.
            Lg, Rg = torch.from_numpy(L / 255.0).to(dev), torch.from_numpy(R / 255.0).to(dev)
            C = stereomatch.cost.SSD(64, kernel_size=3)(Lg, Rg)
            wta = stereomatch.disparity_reduce.WinnerTakesAll()
            variants = {
                "ssd+wta": wta(C),
                "sgm reference": wta(stereomatch.aggregation.Semiglobal(0.1, 0.2)(C, Lg)),
                "sgm true paths": wta(stereomatch.aggregation.Semiglobal(0.1, 0.2, paths="hirschmuller")(C, Lg)),
                "sgm true paths + full min": wta(stereomatch.aggregation.Semiglobal(
                    0.1, 0.2, paths="hirschmuller", full_min=True)(C, Lg)),
            }
            for name, disp in variants.items():
                m = disparity_metrics(disp, gt)
                rows.append((H, W, seed, name, m))
    print("%-28s %9s %8s %8s %8s" % ("variant (mean over scenes)", "rmse", "avgerr", "bad1 %", "bad4 %"))
    for name in ("ssd+wta", "sgm reference", "sgm true paths", "sgm true paths + full min"):
        ms = [m for (_, _, _, n, m) in rows if n == name]
        print("%-28s %9.3f %8.3f %8.2f %8.2f" % (name, np.mean([m["rmse"] for m in ms]),
              np.mean([m["avgerr"] for m in ms]), np.mean([m["bad1"] for m in ms]),
              np.mean([m["bad4"] for m in ms])))


if __name__ == "__main__":
    main()
