"""The ghost-particle halo kernels (border / forward / migrate pack+unpack, csrc/sf_dem_halo.hip) on one GPU:
a single slab whose x-halo goes through the SlabDriver protocol (its own periodic images arrive as "external"
ghosts) must reproduce the engine's internal periodic handling -- and therefore the oracle."""
import numpy as np
import pytest

from sedifoam_amd import synthetic
from tests import dem_cases as dc
import tests.test_dem_gpu as T

pytestmark = pytest.mark.gpu


def _driver(bed, cfg):
    from sedifoam_amd.halo import SlabDriver, HipSlabEngine
    lmp = dc.make_hip(bed, cfg)
    drv = SlabDriver(HipSlabEngine(lmp), None, 0, 1, float(bed["boxlo"][0]), float(bed["boxhi"][0]),
                     periodic_x=True)
    return lmp, drv


@pytest.mark.parametrize("vmax,skin,steps", [(0.01, 0.25e-3, (1, 30)), (0.5, 0.05e-3, (70, 70))])
def test_self_halo_matches_internal_periodic_and_oracle(vmax, skin, steps):
    bed = T._bed((6, 6, 6), periodic=True, seed=77, vmax=vmax)
    cfg = dict(T.BASE, skin=skin)
    cfg["walls"] = T._walls(bed)
    ref = dc.make_hip(bed, cfg)
    orc = dc.make_oracle(bed, cfg)
    lmp, drv = _driver(bed, cfg)
    ref.setup(); orc.setup(); drv.setup()
    a, b = lmp.get_state(), ref.get_state()
    assert dc.rel_err(a["f"], b["f"]) <= 1e-13 and dc.rel_err(a["torque"], b["torque"]) <= 1e-13
    assert lmp.info().nghost == ref.info().nghost
    for n in steps:
        ref.step(n); orc.run(n); drv.step(n)
        a, b, c = lmp.get_state(), ref.get_state(), orc.get()
        assert (a["tag"] == b["tag"]).all()
        for k in ("x", "v", "omega", "f", "torque"):
            assert dc.rel_err(a[k], b[k]) <= 1e-11, k
            assert dc.rel_err(a[k], c[k]) <= 1e-9, k
        ha, hb = lmp.history(), ref.history()
        assert set(ha) == set(hb)
    if vmax > 0.1:
        assert drv.n_rebuilds >= 3      # migration across the periodic face + history carry-over exercised
