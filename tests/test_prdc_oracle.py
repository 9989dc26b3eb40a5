"""CPU: the oracle restatement of prdc.compute_prdc (oracle/fid.py; the package is a non-vendored dependency of the
reference, helpers/metric.py:10,52) on cases with known answers."""
import numpy as np

from oracle import fid as OF


def test_prdc_known_answers():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((200, 16))
    same = OF.compute_prdc(x, x, nearest_k=5)
    assert same["precision"] == 1.0 and same["recall"] == 1.0 and same["coverage"] == 1.0
    # density of a set against itself: every sample lies inside the radius of its k nearest neighbours' ... >= 1 on average
    assert same["density"] > 0.9
    far = OF.compute_prdc(x, x + 100.0, nearest_k=5)
    assert far == {"precision": 0.0, "recall": 0.0, "density": 0.0, "coverage": 0.0}
    # a fake set collapsed onto one real sample: perfect precision, no recall beyond that neighbourhood
    one = np.repeat(x[:1], 50, axis=0) + 1e-6 * rng.standard_normal((50, 16))
    col = OF.compute_prdc(x, one, nearest_k=5)
    assert col["precision"] == 1.0 and col["coverage"] <= 7 / 200 and col["recall"] <= 0.02
    # radii: hand-checkable 1-D case, k = 1: radius = distance to the nearest other point
    r = np.array([[0.0], [1.0], [3.0]])
    f = np.array([[0.4], [2.6], [10.0]])
    out = OF.compute_prdc(r, f, nearest_k=1)
    # real radii: 1, 1, 2; fake radii: 2.2, 2.2, 7.4; d(r, f) = [[.4, 2.6, 10], [.6, 1.6, 9], [2.6, .4, 7]]
    assert out["precision"] == 2 / 3 and out["coverage"] == 1.0 and out["recall"] == 1.0
    assert abs(out["density"] - (2 + 1 + 0) / 3) < 1e-12
