"""Accuracy evaluation over a Middlebury-style dataset directory.

Counterpart of upstream ``workflows/evaluation/workflow.py:24-142`` without flytekit /
stereomideval: every scene is run through the named pipelines (upstream's three: SSD+WTA,
SSD+DP, SSD+SGM+DP) and RMSE, average error and bad-pixel rate against ``disp0.pfm`` are
averaged over the scenes.

    python -m stereomatch.evaluation path/to/middlebury/data
"""
import argparse
from typing import Iterable, List, Optional, Sequence, Tuple

import torch

from .cli_common import create_pipeline
from .data import MiddleburyDataset, disparity_metrics
from .numeric import next_power_of_2

UPSTREAM_CONFIGURATIONS = (("ssd", "wta", None), ("ssd", "dyn", None), ("ssd", "dyn", "sgm"))


def to_grayscale(image: torch.Tensor) -> torch.Tensor:
    """[H, W, 3] (or [H, W]) uint8 -> [H, W] float32 with the ITU-R 601 weights
    torchvision's ``rgb_to_grayscale`` uses upstream (workflow.py:30-33); not rescaled."""
    if image.dim() == 2:
        return image.float()
    rgb = image[..., :3].float()
    return 0.2989 * rgb[..., 0] + 0.587 * rgb[..., 1] + 0.114 * rgb[..., 2]


def predict_disparities(dataset, cost_func: str, disp_func: str, aggr_func: Optional[str] = None,
                        device: str = "cuda:0", **options) -> Iterable[Tuple[int, str, torch.Tensor]]:
    """(index, scene name, int32 disparity map) per scene; the disparity range is the
    scene's ``ndisp`` rounded up to a power of two, as upstream (workflow.py:34)."""
    for idx in range(len(dataset)):
        item = dataset.get_stereo_pair(idx)
        pipeline = create_pipeline(cost_func, disp_func, aggr_func,
                                   max_disparity=next_power_of_2(item["max_disparity"]), **options)
        left = to_grayscale(item["left"]).to(device)
        right = to_grayscale(item["right"]).to(device)
        yield idx, item["stereo_name"], pipeline.estimate(left, right)


def evaluate_dataset(dataset, configurations: Sequence = UPSTREAM_CONFIGURATIONS,
                     device: str = "cuda:0", bad_threshold: float = 2.0, **options) -> List[dict]:
    """One row per configuration: mean RMSE, mean average error, mean bad-pixel rate (%)."""
    rows = []
    for cost_func, disp_func, aggr_func in configurations:
        rmse = avgerr = bad = 0.0
        for idx, _, disparity in predict_disparities(dataset, cost_func, disp_func, aggr_func,
                                                     device, **options):
            truth = dataset.get_ground_truth(idx)["gt_disparity"]
            m = disparity_metrics(disparity, truth, bad_thresholds=(bad_threshold,))
            rmse += m["rmse"]
            avgerr += m["avgerr"]
            bad += m["bad%g" % bad_threshold]
        n = max(len(dataset), 1)
        rows.append({"pipeline": "%s-%s-%s" % (cost_func, disp_func, aggr_func),
                     "rmse": rmse / n, "avgerr": avgerr / n, "bad_pix_error": bad / n})
    return rows


def to_markdown(rows: List[dict]) -> str:
    lines = ["| Alg. pipeline | rmse | avgerr | bad_pix_error |", "|---|---|---|---|"]
    lines += ["| %s | %.4f | %.4f | %.3f |" % (r["pipeline"], r["rmse"], r["avgerr"], r["bad_pix_error"])
              for r in rows]
    return "\n".join(lines)


def main(argv=None):
    parser = argparse.ArgumentParser(description=__doc__,
                                     formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument("dataset_dir")
    parser.add_argument("--max-size", type=int, default=None)
    parser.add_argument("--device", default="cuda:0")
    args = parser.parse_args(argv)
    rows = evaluate_dataset(MiddleburyDataset(args.dataset_dir, args.max_size), device=args.device)
    print(to_markdown(rows))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
