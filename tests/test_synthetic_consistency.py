"""CPU: product-side geometry / synthetic checkpoints == the oracle's independent copy."""
import numpy as np
import pytest

from layout_dm_amd import synthetic as P
from oracle import spec as OS
from oracle import synth as O


@pytest.mark.parametrize("ds", ["rico25", "publaynet"])
def test_same_checkpoint(ds):
    a = P.synth_state_dict(P.SPECS[ds], seed=3, perturb=True)
    b = O.synth_state_dict(OS.SPECS[ds], seed=3, perturb=True)
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert P.SPECS[ds].n_class == OS.SPECS[ds].n_class
    for at in range(5):
        assert np.array_equal(P.SPECS[ds].full_ids(at), OS.SPECS[ds].full_ids(at))


def test_timestep_schedule_matches_oracle():
    from layout_dm_amd.diffusion import timestep_schedule
    from oracle import restatement as R

    for te in (100, 50, 25, 10, 7, 1):
        tm, tp = timestep_schedule(100, te)
        assert tm == R.timestep_list(100, te)
        prev = 100
        for t, p in zip(tm, tp):
            skip = prev - t - 1
            assert p == (t - skip if (skip > 0 and t > skip) else t)
            prev = t
