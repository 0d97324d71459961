"""development helper: step the HIP engine and the oracle side by side through many rebuilds of a fast periodic bed and
print the first sub-step at which forces, history sets or ghost counts part (how the pre_exchange order bug was found)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sedifoam_amd import synthetic
from tests import dem_cases as dc
import tests.test_dem_gpu as T
bed = T._bed((6, 6, 6), periodic=True, seed=99, vmax=0.5)
cfg = dict(T.BASE, skin=0.05e-3); cfg["walls"] = T._walls(bed)
lmp = dc.make_hip(bed, cfg); orc = dc.make_oracle(bed, cfg)
lmp.setup(); orc.setup()
for it in range(40):
    lmp.step(3); orc.run(3)
    a, b = lmp.get_state(), orc.get()
    ha, hb = lmp.history(), orc.history()
    common = sorted(set(ha) & set(hb))
    sa = np.array([ha[k] for k in common]); sb = np.array([hb[k] for k in common])
    print("it %2d builds hip %d orc %d | ef %.2e et %.2e ex %.2e ev %.2e | hist %d/%d common %d esh %.2e ghosts %d/%d" % (
        it, lmp.info().nbuilds, orc.nbuilds, dc.rel_err(a["f"], b["f"]), dc.rel_err(a["torque"], b["torque"]),
        np.abs(a["x"]-b["x"]).max(), dc.rel_err(a["v"], b["v"]), len(ha), len(hb), len(common), dc.rel_err(sa, sb) if len(common) else 0,
        lmp.info().nghost, orc.nghost))
    if dc.rel_err(a["f"], b["f"]) > 1e-6:
        bad = np.argsort(-np.abs(a["f"]-b["f"]).max(axis=1))[:5]
        for i in bad:
            print("   tag", a["tag"][i], "x", a["x"][i], "f hip", a["f"][i], "f orc", b["f"][i])
        missing = sorted(set(hb) - set(ha))[:10]; extra = sorted(set(ha) - set(hb))[:10]
        print("   missing in hip", missing, "extra in hip", extra)
        break
