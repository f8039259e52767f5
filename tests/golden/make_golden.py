#!/usr/bin/env python3
"""Regenerates the fixtures under tests/golden/ from the CPU oracle.

The oracle is first checked against the reference-output digests of SURVEY.md Appendix C
(digests.json); only then are the small full-tensor fixtures written.  Run from the
repository root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402


def check_digests():
    with open(os.path.join(HERE, "digests.json")) as f:
        golden = json.load(f)
    for case, g in golden.items():
        if case.startswith("_"):
            continue
        L, R = oracle.synthetic_pair(g["H"], g["W"], seed=0)
        if "ssd" in g:
            C = oracle.ssd(L, R, g["D"], 7)
            S = oracle.sgm(C, L)
            got = dict(L=oracle.digest16(L), R=oracle.digest16(R), ssd=oracle.digest16(C),
                       ssd_wta=oracle.digest16(oracle.wta(C)), sgm=oracle.digest16(S),
                       ssd_sgm_wta=oracle.digest16(oracle.wta(S)))
        else:
            B = oracle.birchfield(L, R, g["D"])
            S = oracle.sgm(B, L)
            got = dict(birchfield=oracle.digest16(B), sgm=oracle.digest16(S),
                       dp=oracle.digest16(oracle.dp(S)))
        for key, val in got.items():
            assert g[key] == val, (case, key, g[key], val)
        print("pinned:", case)


def write_case(name, H, W, D, k, seed, with_reference_dp):
    L, R = oracle.synthetic_pair(H, W, seed=seed)
    C = oracle.ssd(L, R, D, k)
    B = oracle.birchfield(L, R, D)
    S = oracle.sgm(C, L)
    out = dict(left=L, right=R, D=np.int32(D), k=np.int32(k), ssd=C, birchfield=B, sgm=S,
               wta=oracle.wta(S), dp_corrected=oracle.dp(S, mode=1))
    if with_reference_dp:
        out["dp"] = oracle.dp(S)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: getattr(v, "shape", v) for k, v in out.items()})


def write_oracle_digests():
    """Digests of the ORACLE's outputs (not of the reference's: the survey session did not
    run the reference at this size) for configs 4/5: the 1080p x 256 seed-0 pair through
    SSD(256, k=7) -> SGM(0.1, 0.2) -> WTA.  The oracle itself is pinned by check_digests().
    Takes ~1.5 minutes on one core."""
    H, W, D = 1080, 1920, 256
    L, R = oracle.synthetic_pair(H, W, seed=0)
    C = oracle.ssd(L, R, D, 7)
    S = oracle.sgm(C, L)
    disp = oracle.wta(S)
    out = {"_source": "oracle/stm_oracle.c outputs (sha256, first 16 hex, raw little-endian bytes) "
                      "on the SURVEY 8(d) generator, seed 0; written by tests/golden/make_golden.py",
           "1080p 1080x1920x256": dict(H=H, W=W, D=D, L=oracle.digest16(L), R=oracle.digest16(R),
                                       ssd=oracle.digest16(C), ssd_wta=oracle.digest16(oracle.wta(C)),
                                       sgm=oracle.digest16(S), ssd_sgm_wta=oracle.digest16(disp),
                                       pixels_at_true_disparity=float((disp == 7).mean()))}
    with open(os.path.join(HERE, "oracle_digests.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote oracle_digests.json", out)


if __name__ == "__main__":
    oracle.build()
    check_digests()
    if "--with-1080p" in sys.argv:
        write_oracle_digests()
    write_case("tiny_wide", 24, 40, 16, 7, 0, True)    # W > H: re-walked diagonals (Q6)
    write_case("tiny_tall", 40, 24, 16, 7, 0, False)   # H > W+1: pre-walks; DP reference mode n/a
    write_case("dp_square", 20, 20, 16, 3, 1, True)    # D <= H <= W
