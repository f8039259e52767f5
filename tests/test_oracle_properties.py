"""Independent checks of the oracle: a second, numpy-level restatement for small cases
and the path-coverage numbers SURVEY A.3 (Q6) records for the reference's path table."""
import numpy as np
import pytest


def ssd_numpy(L, R, D, k):
    H, W = L.shape
    out = np.full((H, W, D), np.inf, dtype=np.float32)
    for r in range(H):
        for c in range(W):
            for d in range(min(c + 1, D)):
                s = np.float32(0)
                for kr in range(max(r - k, 0), min(r + k, H)):
                    for kc in range(max(c - k, d), min(c + k, W)):
                        t = np.float32(L[kr, kc] - R[kr, kc - d])
                        s = np.float32(s + np.float32(t * t))
                out[r, c, d] = s
    return out


def test_ssd_against_numpy_loops(oracle):
    L, R = oracle.synthetic_pair(9, 13, seed=4, kind="unit")
    got = oracle.ssd(L, R, 6, 2)
    assert np.array_equal(got.view(np.uint32), ssd_numpy(L, R, 6, 2).view(np.uint32))


def test_wta_is_first_minimum(oracle):
    rng = np.random.default_rng(0)
    vol = rng.integers(0, 3, size=(12, 14, 32)).astype(np.float32)   # many ties
    vol[rng.random(vol.shape) < 0.1] = np.inf
    assert np.array_equal(oracle.wta(vol), np.argmin(vol, axis=2).astype(np.int32))


def sgm_numpy(C, I, p1, p2, paths):
    """Literal path-table walk in float32 numpy (Appendix A.3)."""
    H, W, D = C.shape
    S = np.zeros_like(C)
    p1 = np.float32(p1)
    p2 = np.float32(p2)
    inf = np.float32(np.inf)

    def run(x, y, dx, dy, n):
        prev = C[y, x].copy()
        S[y, x] += prev
        pi = I[y, x]
        for _ in range(1, n):
            m = prev[:D - 1].min() if D > 1 else prev[0]
            x += dx
            y += dy
            with np.errstate(divide="ignore", invalid="ignore"):
                q = np.float32(p2 / np.abs(np.float32(I[y, x] - pi)))
            p2a = q if p1 < q else p1
            pi = I[y, x]
            lo = np.concatenate([[inf], prev[:-1]]) + p1
            hi = np.concatenate([prev[1:], [inf]]) + p1
            t = prev.copy()
            t = np.where(lo < t, lo, t)
            t = np.where(hi < t, hi, t)
            far = np.float32(m + p2a)
            t = np.where(far < t, far, t)
            cur = ((C[y, x] + t).astype(np.float32) - m).astype(np.float32)
            S[y, x] += cur
            prev = cur

    for sx, sy, ex, ey, dx, dy, n in paths:
        run(int(sx), int(sy), int(dx), int(dy), int(n))
        run(int(ex), int(ey), -int(dx), -int(dy), int(n))
    return S


@pytest.mark.parametrize("H,W,D", [(7, 7, 5), (6, 11, 8), (11, 6, 8), (9, 14, 4)])
def test_sgm_against_numpy_walk(oracle, H, W, D):
    rng = np.random.default_rng(H * W)
    C = (rng.random((H, W, D), dtype=np.float32) * 20).astype(np.float32)
    C[..., D - 1] *= 0.01      # minimum often at the last disparity (Q5)
    I = rng.integers(0, 4, size=(H, W)).astype(np.float32)
    want = sgm_numpy(C, I, 0.1, 0.2, oracle.sgm_paths(W, H))
    got = oracle.sgm(C, I, 0.1, 0.2)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("W,H,unvisited,repeats", [(450, 375, 28125, 76), (1920, 1080, 907200, 841),
                                                   (512, 256, 65536, 257), (300, 300, 0, 1)])
def test_path_table_coverage_matches_survey(oracle, W, H, unvisited, repeats):
    """pixels that get no up-left pass, and the pass count of the re-walked
    diagonal, for the reference's inverse passes of the (1,1) family."""
    paths = oracle.sgm_paths(W, H)
    assert len(paths) == H + W + 2 * (W + H - 1)
    visits = np.zeros((H, W), dtype=np.int32)
    for sx, sy, ex, ey, dx, dy, n in paths:
        if (dx, dy) == (1, 1):
            xs = ex - np.arange(n)
            ys = ey - np.arange(n)
            assert xs.min() >= 0 and ys.min() >= 0 and xs.max() < W and ys.max() < H
            np.add.at(visits, (ys, xs), 1)
    assert int((visits == 0).sum()) == unvisited
    assert int(visits.max()) == repeats


def test_dp_corrected_against_numpy(oracle):
    rng = np.random.default_rng(3)
    H, W, D = 5, 9, 6
    vol = rng.integers(0, 5, size=(H, W, D)).astype(np.float32)
    disp, path, final = oracle.dp(vol, mode=1, return_aux=True)
    for row in range(H):
        prev = vol[row, 0].copy()
        for col in range(1, W):
            for d in range(D):
                c1 = prev[d - 1] if d > 0 else np.inf
                c2 = prev[d]
                c3 = prev[d + 1] if d < D - 1 else np.inf
                if c1 < c2 and c1 < c3:
                    m, direction = c1, -1
                elif c2 < c3:
                    m, direction = c2, 0
                else:
                    m, direction = c3, 1
                assert path[row, col, d] == direction
                prev[d] = np.float32(vol[row, col, d] + m)
        assert np.array_equal(prev, final[row])
        cur = int(np.argmin(prev))
        assert disp[row, W - 1] == cur
        for col in range(W - 2, -1, -1):
            cur = min(max(cur + int(path[row, col, cur]), 0), D - 1)
            assert disp[row, col] == cur
