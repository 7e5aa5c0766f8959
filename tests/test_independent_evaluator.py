"""The C oracle against a second opinion: tests/independent_evaluator.py (numpy, written from
SURVEY.md Appendix B and the reference's kernel source, not from the oracle) on the six golden
cases and on seeded small cases of both models, invalid_disparity -1 / 0: complete cost_table,
index_table and Sections bit for bit.  Cheap risk reduction against a formula mis-transcribed in
the oracle -- it does not change the pin status of the oracle."""
import os

import numpy as np
import pytest

import helpers
import independent_evaluator as ie
from instance_stixels_amd.config import SECTION_DTYPE
from test_oracle_properties import load_golden, GOLDEN


def _compare(p, got, ref_ct, ref_it, ref_sections):
    ct, it, secs = got
    assert np.array_equal(ct.view(np.uint32), ref_ct.view(np.uint32)), \
        f"cost_table differs in {int((ct.view(np.uint32) != ref_ct.view(np.uint32)).sum())} entries"
    written = ref_it >= 0
    assert np.array_equal(it[written], ref_it[written])
    assert np.array_equal(it >= 0, written)
    for c in range(p.cols):
        n = helpers.n_sections(ref_sections[c])
        assert len(secs[c]) == n, (c, len(secs[c]), n)
        mine = np.zeros(n, SECTION_DTYPE)
        for i, s in enumerate(secs[c]):
            mine[i] = s
        assert np.array_equal(mine.view(np.int32), ref_sections[c][:n].view(np.int32)), c


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_independent_evaluator_equals_oracle_on_golden_cases(path):
    g = load_golden(path)
    got = ie.evaluate(g["params"], g["joined"], g["segmentation"], g["gf"], g["ng"], g["ig"], int(g["vhor"]),
                      g["lut"], g["odr"], bool(g["pairwise"]))
    _compare(g["params"], got, g["cost_table"], g["index_table"], g["sections"])


@pytest.mark.parametrize("preset,rows,cols,D,ov,seed", [
    ("drn_d_22_unary", 64, 64, 32, {}, 1), ("drn_d_38_pairwise", 64, 64, 32, {}, 2),
    ("drn_d_22_pairwise", 64, 32, 16, dict(invalid_disparity=0.0), 3),
    ("drn_d_38_unary", 56, 32, 32, dict(invalid_disparity=0.0), 4),
    ("drn_d_38_pairwise", 40, 64, 16, dict(prior_weight=0.5, epsilon=1.0), 5)])
def test_independent_evaluator_equals_oracle_on_seeded_cases(preset, rows, cols, D, ov, seed):
    case = helpers.build_case(preset, rows, cols, D, seed=seed, **ov)
    ref = helpers.run_oracle(case)
    p = case["params"]
    p.vhor = int(case["vhor"][0])
    got = ie.evaluate(p, ref["joined"], case["segmentation"][0], case["gf"][0], case["ng"][0], case["ig"][0],
                      int(case["vhor"][0]), case["lut"], case["odr"], bool(case["cfg"].pairwise))
    _compare(p, got, ref["cost_table"], ref["index_table"], ref["sections"])
