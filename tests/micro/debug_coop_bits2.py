"""development: the first sub-step at which the half-wave gather (policy 2) and the plain gather (policy 0) differ"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
os.environ["SF_LPA"] = "1"
os.environ["SF_TOUCH_PREFETCH"] = "0"
from tests import dem_cases as dc
from tests.test_dem_gpu import _bed, _walls, BASE
bed = _bed((7, 5, 6), periodic=False, seed=21, vmax=0.6)
cfg = dict(BASE, skin=0.05e-3, walls=_walls(bed))
if os.environ.get("NO_SLIDING"):
    cfg["xmu"] = 1.0e4   # (no pair ever reaches the Coulomb limit)
eng = {}
for policy in ("0", "2"):
    os.environ["SF_NT_POLICY"] = policy
    eng[policy] = dc.make_hip(bed, cfg)
    eng[policy].setup()
shown = 0
for k in range(1, 60):
    for p in eng.values():
        p.step(1)
    a, b = eng["0"].get_state(), eng["2"].get_state()
    nd = {q: int(np.sum(np.any(a[q] != b[q], axis=-1))) for q in ("x", "v", "omega", "f", "torque")}
    if any(nd.values()):
        idx = np.nonzero(np.any(a["f"] != b["f"], axis=-1) | np.any(a["torque"] != b["torque"], axis=-1))[0]
        print("step", k, "builds", eng["0"].info().nbuilds, eng["2"].info().nbuilds, "differing", nd)
        for i in idx[:6]:
            print("   atom tag", a["tag"][i], "f", a["f"][i], b["f"][i] - a["f"][i], "torque diff", b["torque"][i] - a["torque"][i])
        ha, hb = eng["0"].history(), eng["2"].history()
        bad = [q for q in ha if q in hb and not np.array_equal(ha[q], hb[q])]
        print("   history pairs differing", len(bad), "of", len(ha), bad[:4], [ (ha[q], hb[q]) for q in bad[:2]])
        shown += 1
        if shown >= 3:
            break
print("steps done", k, "differences shown", shown)
