"""Pins oracle/fusion.py against run.get_c_expr_db_pred / data.utils.get_compound_expression of the reference."""
import numpy as np

from oracle import fusion as of


def test_known_answer_weights(golden):
    g = golden("fusion")
    # run.py:316-344 is get_weights_matrices.py:51-59 (first 7 rows) transposed -- the only numeric fixture
    # the reference holds for this path
    np.testing.assert_array_equal(g["weights_3"][:7].T, g["weights_av_1"])
    np.testing.assert_array_equal(np.array(of.WEIGHTS_AV_1), g["weights_av_1"])


def test_fuse_matches_reference(golden):
    g = golden("fusion")
    for c in range(int(g["n_cases"])):
        stat, dyn = g[f"c{c}_stat"], g[f"c{c}_dyn"]
        rows, frames = g[f"c{c}_aud_rows"], g[f"c{c}_aud_frames"]
        for wname, w1 in (("w", of.WEIGHTS_AV_1), ("none", None)):
            for cwt in (False, True):
                for cm in (False, True):
                    key = f"c{c}_{wname}_{int(cwt)}{int(cm)}"
                    prob, am = of.fuse(stat, dyn, rows, frames, w1, (1, 1, 1), cwt, cm)
                    assert prob.dtype == np.float64 == g[key + "_prob"].dtype
                    np.testing.assert_allclose(prob, g[key + "_prob"], rtol=0, atol=1e-7)
                    np.testing.assert_array_equal(am, g[key + "_argmax"])


def test_softmax_rows():
    x = np.array([[1000.0, 1000.0, -1000.0]], dtype=np.float32)
    s = of.softmax(x)
    assert s.dtype == np.float32 and np.allclose(s, [[0.5, 0.5, 0.0]])
