"""Middlebury-2021-style dataset directory reader (counterpart of upstream
``stereomatch/data.py:26-93``) without the cv2 / natsort dependencies: PFM ground truth is
parsed here, scene directories are ordered naturally (scene2 before scene10)."""
import re
from pathlib import Path
from typing import Optional

import numpy as np
import torch
from PIL import Image


def read_pfm(path) -> np.ndarray:
    """A PFM file as a top-down float32 array ([H, W] for ``Pf``, [H, W, 3] for ``PF``),
    the orientation ``cv2.imread(..., IMREAD_UNCHANGED)`` returns upstream (data.py:86)."""
    with open(path, "rb") as f:
        kind = f.readline().strip()
        if kind not in (b"Pf", b"PF"):
            raise RuntimeError("read_pfm: %s is not a PFM file" % path)
        dims = f.readline()
        while dims.startswith(b"#"):
            dims = f.readline()
        width, height = (int(v) for v in dims.split())
        scale = float(f.readline().strip())
        channels = 3 if kind == b"PF" else 1
        data = np.frombuffer(f.read(width * height * channels * 4),
                             dtype="<f4" if scale < 0 else ">f4")
    shape = (height, width, 3) if channels == 3 else (height, width)
    return np.ascontiguousarray(np.flipud(data.reshape(shape)).astype(np.float32))


def write_pfm(path, array: np.ndarray) -> None:
    """Little-endian single-channel PFM (used by the tests and for exporting results)."""
    array = np.asarray(array, dtype=np.float32)
    with open(path, "wb") as f:
        f.write(b"Pf\n%d %d\n-1.0\n" % (array.shape[1], array.shape[0]))
        f.write(np.flipud(array).astype("<f4").tobytes())


def _natural_key(path: Path):
    return [int(tok) if tok.isdigit() else tok.lower() for tok in re.split(r"(\d+)", path.name)]


def _parse_calib(path):
    props = {}
    with open(path, "r", encoding="ascii") as f:
        for line in f:
            if "=" in line:
                name, value = line.split("=", 1)
                props[name.strip()] = value.strip()
    return dict(width=int(props["width"]), height=int(props["height"]), ndisp=int(props["ndisp"]))


class MiddleburyDataset:
    """Scenes are sub-directories holding ``im0.png``, ``im1.png``, ``disp0.pfm`` and
    ``calib.txt``.  ``dataset[i]`` returns a dict with ``stereo_name``, ``left``, ``right``
    (uint8 tensors as stored), ``gt_disparity`` (float32, ``inf`` where unknown) and
    ``max_disparity`` (``ndisp`` of the calibration file), like upstream."""

    def __init__(self, dataset_dir: str, max_size: Optional[int] = None):
        dataset_dir = Path(dataset_dir)
        if not dataset_dir.is_dir():
            raise RuntimeError("MiddleburyDataset: %s must be a directory" % dataset_dir)
        scenes = sorted((d for d in dataset_dir.iterdir() if d.is_dir()), key=_natural_key)
        if max_size is not None:
            scenes = scenes[:max_size]
        self.images = [(d / "im0.png", d / "im1.png") for d in scenes]
        self.disps = [(d / "disp0.pfm", d / "disp1.pfm") for d in scenes]
        self.calibs = [_parse_calib(d / "calib.txt") for d in scenes]

    def __len__(self):
        return len(self.images)

    def __getitem__(self, idx: int):
        item = self.get_stereo_pair(idx)
        item.update(self.get_ground_truth(idx))
        return item

    def get_stereo_pair(self, idx: int):
        left_path, right_path = self.images[idx]
        return dict(stereo_name=left_path.parent.name,
                    left=torch.from_numpy(np.array(Image.open(left_path))),
                    right=torch.from_numpy(np.array(Image.open(right_path))),
                    max_disparity=self.calibs[idx]["ndisp"])

    def get_ground_truth(self, idx: int):
        disp_path = self.disps[idx][0]
        return dict(stereo_name=disp_path.parent.name,
                    gt_disparity=torch.from_numpy(read_pfm(disp_path)),
                    max_disparity=self.calibs[idx]["ndisp"])


def disparity_metrics(estimate, ground_truth, bad_thresholds=(0.5, 1.0, 2.0, 4.0)):
    """RMSE, average absolute error and bad-pixel percentages over the pixels with a finite
    ground truth -- the figures upstream's evaluation workflow collects through
    ``stereomideval`` (``workflows/evaluation/workflow.py:74-105``)."""
    est = torch.as_tensor(estimate).to(torch.float64).cpu()
    gt = torch.as_tensor(ground_truth).to(torch.float64).cpu()
    valid = torch.isfinite(gt)
    err = (est[valid] - gt[valid]).abs()
    out = {"valid_pixels": int(valid.sum()),
           "rmse": float(torch.sqrt((err * err).mean())) if err.numel() else float("nan"),
           "avgerr": float(err.mean()) if err.numel() else float("nan")}
    for t in bad_thresholds:
        out["bad%g" % t] = float((err > t).double().mean() * 100.0) if err.numel() else float("nan")
    return out
