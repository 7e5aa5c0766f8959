"""Writes tests/golden/*.npz: small kernel-boundary vectors (inputs + expected outputs).

PROVENANCE: the expected outputs are produced by THIS repository's CPU oracle
(oracle/stixels_oracle.c), not by the reference: the reference is CUDA and cannot be run in this
image, and its own tests hold no vector for this path.  They are regression
vectors that pin today's oracle + HIP behaviour, so that any later change of either is caught.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402

CASES = {
    "unary_64x64x32": ("drn_d_22_unary", 64, 64, 32, {}),
    "pairwise_64x64x32": ("drn_d_38_pairwise", 64, 64, 32, {}),
    "unary_invalid0_128x64x32": ("drn_d_38_unary", 128, 64, 32, dict(invalid_disparity=0.0)),
    "pairwise_invalid0_128x64x32": ("drn_d_22_pairwise", 128, 64, 32, dict(invalid_disparity=0.0)),
    "disparity_only_unary_128x64x64": ("disparity_only_unary", 128, 64, 64, {}),
    "disparity_only_pairwise_128x64x64": ("disparity_only_pairwise", 128, 64, 64, {}),
}

if __name__ == "__main__":
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, (preset, rows, cols, D, ov) in CASES.items():
        case = helpers.build_case(preset, rows, cols, D, seed=2024, **ov)
        ref = helpers.run_oracle(case)
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"),
            params=np.frombuffer(bytes(case["params"]), np.uint8), lut=case["lut"],
            odr=case["odr"], gf=case["gf"][0], ng=case["ng"][0], ig=case["ig"][0],
            vhor=case["vhor"][0], pairwise=int(case["cfg"].pairwise),
            disparity=case["disparity"][0], median_join=int(case["cfg"].median_join),
            segmentation=case["segmentation"][0], joined=ref["joined"],
            sections=ref["sections"], cost_table=ref["cost_table"],
            index_table=ref["index_table"], inst_per_class=ref["inst_per_class"],
            inst_indices=ref["inst_indices"], inst_centerofmass=ref["inst_centerofmass"],
            inst_core=ref["inst_core"])
        print("wrote", name, os.path.getsize(os.path.join(out_dir, name + ".npz")), "bytes")
