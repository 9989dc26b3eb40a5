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


def test_prdc_against_an_independent_neighbour_search():
    """The same four numbers from scikit-learn's exact k-NN search (radii) and a plain double loop over thresholded
    distances — nothing shared with the restatement's pairwise-matrix / partition route (Naeem et al. 2020, eqs. 3-5:
    precision / recall = share of samples inside at least one k-NN ball of the other set, density = mean number of real
    balls a fake sample falls into / k, coverage = share of real samples whose ball holds a fake sample)."""
    from sklearn.neighbors import NearestNeighbors

    rng = np.random.default_rng(3)
    for nr, nf, d, k in ((120, 90, 8, 5), (64, 333, 48, 3), (257, 257, 24, 7)):
        real = rng.standard_normal((nr, d))
        fake = 0.8 * rng.standard_normal((nf, d)) + 0.3
        rad_r = NearestNeighbors(n_neighbors=k + 1, algorithm="brute").fit(real).kneighbors(real)[0][:, k]
        rad_f = NearestNeighbors(n_neighbors=k + 1, algorithm="brute").fit(fake).kneighbors(fake)[0][:, k]
        dist = np.sqrt(((real[:, None, :] - fake[None, :, :]) ** 2).sum(-1))
        in_real = dist < rad_r[:, None]   # fake j inside the ball of real i
        in_fake = dist < rad_f[None, :]   # real i inside the ball of fake j
        want = {"precision": in_real.any(0).mean(), "recall": in_fake.any(1).mean(),
                "density": in_real.sum(0).mean() / k, "coverage": (dist.min(1) < rad_r).mean()}
        got = OF.compute_prdc(real, fake, nearest_k=k)
        for key in want:
            assert abs(got[key] - want[key]) < 1e-12, (key, got[key], want[key])
