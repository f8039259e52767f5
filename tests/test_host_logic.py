"""Host-side behaviour of the stereomatch package that needs no GPU: numeric helpers,
output-buffer policy, Pipeline composition, argument checks."""
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def test_next_power_of_2_table():
    """The reference's only known-answer test (tests/test_numeric.py:7-17)."""
    from stereomatch.numeric import is_power_of_two, next_power_of_2
    assert [next_power_of_2(n) for n in (0, 1, 2, 3, 4, 50, 127)] == [1, 1, 2, 4, 4, 64, 128]
    assert next_power_of_2(128) == 128 and next_power_of_2(129) == 256
    assert [is_power_of_two(n) for n in (0, 1, 2, 3, 64, 100)] == [False, True, True, False, True, False]


def test_output_tensor_reuse_policy():
    """Same object back when shape, dtype and device match; a fresh tensor otherwise
    (upstream _common.py:9-53)."""
    from stereomatch._common import output_tensor
    cpu = torch.device("cpu")
    buf = torch.empty(3, 4, 5)
    assert output_tensor((3, 4, 5), torch.float32, cpu, reuse=buf) is buf
    assert output_tensor((3, 4, 6), torch.float32, cpu, reuse=buf) is not buf
    assert output_tensor((3, 4, 5), torch.int32, cpu, reuse=buf).dtype == torch.int32
    assert output_tensor((2, 2), torch.float32, cpu).shape == (2, 2)
    assert float(output_tensor((2, 2), torch.float32, cpu, zero_fresh=True).abs().sum()) == 0.0


def test_ops_refuse_cpu_tensors_loudly():
    """No CPU fallback in the product path."""
    import stereomatch
    img = torch.zeros(8, 8)
    vol = torch.zeros(8, 8, 8)
    for call in (lambda: stereomatch.cost.SSD(8)(img, img),
                 lambda: stereomatch.cost.Birchfield(8)(img, img),
                 lambda: stereomatch.aggregation.Semiglobal()(vol, img),
                 lambda: stereomatch.disparity_reduce.WinnerTakesAll()(vol),
                 lambda: stereomatch.disparity_reduce.DynamicProgramming()(vol)):
        with pytest.raises(RuntimeError, match="Expected a gpu tensor"):
            call()
    with pytest.raises(RuntimeError, match="texture"):
        stereomatch.cost.SSDTexture(8)
    with pytest.raises(ValueError):
        stereomatch.cost.SSD(8, mode="fast")
    with pytest.raises(ValueError):
        stereomatch.disparity_reduce.DynamicProgramming(mode="x")


def test_constructor_defaults_match_reference():
    import stereomatch
    ssd = stereomatch.cost.SSD(32)
    assert (ssd.max_disparity, ssd.kernel_size, ssd.cost_volume_dtype) == (32, 7, torch.float32)
    sgm = stereomatch.aggregation.Semiglobal()
    assert (sgm.penalty1, sgm.penalty2) == (0.1, 0.2)
    assert stereomatch.cost.Birchfield(16).max_disparity == 16


class _OracleOps:
    """Oracle-backed callables with the op wrappers' signatures (CPU, test only)."""

    def __init__(self, oracle, D):
        self.oracle, self.D = oracle, D
        self.calls = []

    def cost(self, left, right, cost_volume=None):
        from stereomatch._common import output_tensor
        out = output_tensor((left.shape[0], left.shape[1], self.D), torch.float32, left.device,
                            reuse=cost_volume)
        out.copy_(torch.from_numpy(self.oracle.ssd(left.numpy(), right.numpy(), self.D, 7)))
        self.calls.append(("cost", cost_volume is out))
        return out

    def aggregation(self, volume, left, sga_volume=None):
        from stereomatch._common import output_tensor
        out = output_tensor(volume.shape, torch.float32, volume.device, reuse=sga_volume)
        out.copy_(torch.from_numpy(self.oracle.sgm(volume.numpy(), left.numpy())))
        self.calls.append(("aggregation", sga_volume is out))
        return out

    def reduce(self, volume, disparity_img=None):
        from stereomatch._common import output_tensor
        out = output_tensor(volume.shape[:2], torch.int32, volume.device, reuse=disparity_img)
        out.copy_(torch.from_numpy(self.oracle.wta(volume.numpy())))
        self.calls.append(("reduce", disparity_img is out))
        return out


def test_pipeline_plumbing_config1_on_cpu(oracle):
    """BASELINE config 1: Teddy-sized SSD(64) + WTA through Pipeline on the CPU.  The ops
    here are oracle-backed stand-ins (the product ops are GPU-only); what is under test is
    the composition and buffer reuse of stereomatch.Pipeline (upstream pipeline.py:69-94,
    tests/test_pipeline.py:15-37)."""
    import stereomatch
    with open(os.path.join(HERE, "golden", "digests.json")) as f:
        g = json.load(f)["teddy 375x450x64"]
    L, R = oracle.synthetic_pair(g["H"], g["W"], seed=0)
    ops = _OracleOps(oracle, g["D"])
    pipe = stereomatch.Pipeline(ops.cost, ops.reduce)
    Lt, Rt = torch.from_numpy(L), torch.from_numpy(R)
    d1 = pipe.estimate(Lt, Rt)
    assert oracle.digest16(d1.numpy()) == g["ssd_wta"]
    d2 = pipe.estimate(Lt, Rt)
    assert d2 is d1 and oracle.digest16(d2.numpy()) == g["ssd_wta"]
    assert ops.calls == [("cost", False), ("reduce", False), ("cost", True), ("reduce", True)]


def test_pipeline_with_aggregation_reuses_three_buffers(oracle):
    import stereomatch
    L, R = oracle.synthetic_pair(24, 40, seed=0)
    ops = _OracleOps(oracle, 16)
    pipe = stereomatch.Pipeline(ops.cost, ops.reduce, aggregation=ops.aggregation)
    Lt, Rt = torch.from_numpy(L), torch.from_numpy(R)
    first = pipe.estimate(Lt, Rt).clone()
    again = pipe.estimate(Lt, Rt)
    assert torch.equal(first, again)
    assert oracle.digest16(again.numpy()) == "b3a8123344038b43"     # SURVEY App. C, tiny-wide
    assert [c[1] for c in ops.calls] == [False] * 3 + [True] * 3


def test_pair_shard_partitions():
    from stereomatch.distributed import pair_shard
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [pair_shard(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sorted(sizes, reverse=True) == sizes
    assert [pair_shard(64, r, 8) for r in (0, 7)] == [(0, 8), (56, 64)]     # config 5
    with pytest.raises(ValueError):
        pair_shard(4, 2, 2)


def test_create_pipeline_registry_matches_reference_names():
    """Upstream cli_common.py:9-52 (the texture cost is not offered on CDNA4)."""
    import stereomatch
    from stereomatch.cli_common import (AGGREGATION_METHODS, COST_METHODS, DISPARITY_METHODS,
                                        create_pipeline)
    assert sorted(COST_METHODS) == ["birchfield", "ssd"]
    assert sorted(AGGREGATION_METHODS) == ["sgm"] and sorted(DISPARITY_METHODS) == ["dyn", "wta"]
    pipe = create_pipeline("ssd", "wta", "sgm", max_disparity=64)
    assert isinstance(pipe.cost, stereomatch.cost.SSD) and pipe.cost.max_disparity == 64
    assert isinstance(pipe.aggregation, stereomatch.aggregation.Semiglobal)
    assert isinstance(pipe.disparity_reduce, stereomatch.disparity_reduce.WinnerTakesAll)
    assert create_pipeline("birchfield", "dyn", None).aggregation is None
    assert create_pipeline("ssd", "wta").cost.max_disparity == 32      # upstream default


def test_middlebury_reader_and_metrics(tmp_path):
    """Upstream data.py:26-93 (its own test is skipped, tests/test_data.py:9): a synthetic
    scene tree with PNG pairs, PFM ground truth and calib.txt."""
    from PIL import Image
    from stereomatch.data import MiddleburyDataset, disparity_metrics, read_pfm, write_pfm
    rng = np.random.default_rng(0)
    truth = {}
    for name in ("scene10", "scene2", "artroom1"):
        d = tmp_path / name
        d.mkdir()
        left = rng.integers(0, 256, size=(12, 20, 3), dtype=np.uint8)
        right = rng.integers(0, 256, size=(12, 20, 3), dtype=np.uint8)
        gt = rng.random((12, 20), dtype=np.float32) * 30
        gt[0, :3] = np.inf
        Image.fromarray(left).save(d / "im0.png")
        Image.fromarray(right).save(d / "im1.png")
        write_pfm(d / "disp0.pfm", gt)
        (d / "calib.txt").write_text("cam0=[1 0 0; 0 1 0; 0 0 1]\nwidth=20\nheight=12\nndisp=48\n")
        truth[name] = (left, right, gt)
    (tmp_path / "README.txt").write_text("not a scene")
    ds = MiddleburyDataset(str(tmp_path))
    assert [ds[i]["stereo_name"] for i in range(len(ds))] == ["artroom1", "scene2", "scene10"]
    item = ds[1]
    left, right, gt = truth["scene2"]
    assert item["max_disparity"] == 48
    assert np.array_equal(item["left"].numpy(), left) and np.array_equal(item["right"].numpy(), right)
    assert np.array_equal(item["gt_disparity"].numpy(), gt)          # top-down, inf kept
    assert len(MiddleburyDataset(str(tmp_path), max_size=2)) == 2
    assert np.array_equal(read_pfm(tmp_path / "scene2" / "disp0.pfm"), gt)
    est = np.where(np.isfinite(gt), gt, 0) + 1.5
    m = disparity_metrics(est, gt)
    assert m["valid_pixels"] == 12 * 20 - 3
    assert abs(m["rmse"] - 1.5) < 1e-6 and abs(m["avgerr"] - 1.5) < 1e-6
    assert m["bad1"] == 100.0 and m["bad2"] == 0.0
    with pytest.raises(RuntimeError):
        MiddleburyDataset(str(tmp_path / "nope"))


def test_upstream_named_helpers_and_texture_placeholder():
    import stereomatch
    from stereomatch._common import empty_tensor, zeros_tensor_like
    buf = torch.ones(2, 3)
    assert empty_tensor(2, 3, reuse_tensor=buf) is buf
    assert empty_tensor(2, 4, dtype=torch.int32).shape == (2, 4)
    z = zeros_tensor_like(torch.ones(2, 3), reuse_tensor=buf)
    assert z is buf and float(buf.sum()) == 0.0
    assert zeros_tensor_like(torch.ones(4)).shape == (4,)
    with pytest.raises(RuntimeError, match="texture"):
        stereomatch.cuda_texture.CUDATexture()
