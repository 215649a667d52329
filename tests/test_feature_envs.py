"""Feature-vector envs HarvestFeatures / CleanupFeatures (BASELINE config 0): the CPU oracle and the
HIP engine against fixtures produced by the reference itself (tests/golden/feat_*.npz)."""
import numpy as np
import pytest

import golden_check as gc

FEAT = gc.fixtures("feat_")


@pytest.mark.parametrize("name", FEAT)
def test_oracle_feature_golden(name):
    from oracle.pyoracle import Oracle
    g = gc.load(name)
    kind, n, kw = gc.feat_kwargs(g)
    orc = Oracle(kind, 2, n, **kw)
    gc.replay_feat(g, orc, env=1)
    orc.close()
