"""development: bitwise comparison of the sub-step kernel's cache-policy variants (0, 3: plain gather; 1, 2: half-wave gather)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
os.environ["SF_LPA"] = "1"
os.environ["SF_TOUCH_PREFETCH"] = "0"
from tests import dem_cases as dc
from tests.test_dem_gpu import _bed, _walls, BASE
bed = _bed((7, 5, 6), periodic=False, seed=21, vmax=0.6)
cfg = dict(BASE, skin=0.05e-3, walls=_walls(bed))
outs = {}
for policy in ("0", "3", "1", "2"):
    os.environ["SF_NT_POLICY"] = policy
    for nsteps in ((1,), (40, 25, 40)):
        lmp = dc.make_hip(bed, cfg)
        lmp.setup()
        for n in nsteps:
            lmp.step(n)
        outs[(policy, nsteps)] = (lmp.get_state(), lmp.info().nbuilds)
for nsteps in ((1,), (40, 25, 40)):
    ref = outs[("0", nsteps)]
    for policy in ("3", "1", "2"):
        o = outs[(policy, nsteps)]
        d = {k: float(np.max(np.abs(o[0][k] - ref[0][k])) / (np.max(np.abs(ref[0][k])) + 1e-300)) for k in ("x", "v", "omega", "f", "torque")}
        nd = {k: int(np.sum(np.any(o[0][k] != ref[0][k], axis=-1))) for k in ("x", "v", "omega", "f", "torque")}
        print("steps", nsteps, "policy", policy, "builds", o[1], ref[1], "max rel diff", d, "atoms differing", nd)
