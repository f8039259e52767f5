#!/usr/bin/env python
"""Disparity map of a rectified stereo pair from the shell (counterpart of upstream
``stm-image``, ``stereomatch/cli_image.py:34-100``, minus the matplotlib display).

    python -m stereomatch.cli_image left.png right.png 128 disparity.png -am sgm

Images are read as 8-bit grayscale and fed as float32 *without* scaling, as upstream
(``cli_image.py:70-75``).  The output is a 16-bit PNG of the disparities (or ``.npy``
for the raw int32 map); ``--scale`` stretches them to the full 16-bit range for viewing.
"""
import argparse

import numpy as np
import torch
from PIL import Image

from .cli_common import AGGREGATION_METHODS, COST_METHODS, DISPARITY_METHODS, create_pipeline


def estimate_files(left_path, right_path, max_disparity, cost_method="ssd",
                   aggregation_method=None, disparity_method="wta", device="cuda:0"):
    """int32 disparity map (numpy) of the image pair at the given paths."""
    pipeline = create_pipeline(cost_method, disparity_method, aggregation_method, max_disparity)
    left = torch.from_numpy(np.array(Image.open(left_path).convert("L"))).float().to(device)
    right = torch.from_numpy(np.array(Image.open(right_path).convert("L"))).float().to(device)
    return pipeline.estimate(left, right).cpu().numpy()


def main(argv=None):
    parser = argparse.ArgumentParser(description=__doc__,
                                     formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument("left_image", metavar="left-image", help="Left image")
    parser.add_argument("right_image", metavar="right-image", help="Right image")
    parser.add_argument("max_disparity", metavar="max-disparity", type=int,
                        help="Maximum disparity for stereo matching.")
    parser.add_argument("output_depthmap", metavar="output-depthmap",
                        help="Output file of the disparity map (.png or .npy).")
    parser.add_argument("-cm", "--cost-method", choices=COST_METHODS.keys(), default="ssd")
    parser.add_argument("-am", "--aggregation-method", choices=AGGREGATION_METHODS.keys(),
                        default=None)
    parser.add_argument("-dm", "--disparity-method", choices=DISPARITY_METHODS.keys(), default="wta")
    parser.add_argument("-c", "--cuda-on", action="store_true",
                        help="Accepted for compatibility; this build always runs on the GPU.")
    parser.add_argument("--device", default="cuda:0")
    parser.add_argument("--scale", action="store_true",
                        help="Stretch disparities to the full 16-bit range in the PNG.")
    args = parser.parse_args(argv)
    disp = estimate_files(args.left_image, args.right_image, args.max_disparity, args.cost_method,
                          args.aggregation_method, args.disparity_method, args.device)
    if args.output_depthmap.endswith(".npy"):
        np.save(args.output_depthmap, disp)
    else:
        out = disp.astype(np.uint16)
        if args.scale and args.max_disparity > 1:
            out = (disp.astype(np.float64) * (65535.0 / (args.max_disparity - 1))).astype(np.uint16)
        Image.fromarray(out).save(args.output_depthmap)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
