"""Seeded differential campaign, HIP engine against the oracle (development helper, GPU box):
random small beds -- size, periodicity, packing, polydispersity, pair style, friction, damping, skin, speeds, frozen
layers in both fix orders, cohesion, lubrication (lubricate/poly over the granular style) -- each stepped through several neighbour rebuilds and compared
after every leg (x 1e-9 d, v / omega 1e-9, f / torque 1e-8 of max, same rebuild count, same contact set).
usage: python tests/fuzz_dem.py [first_seed] [cases]"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sedifoam_amd import synthetic   # noqa: E402
from tests import dem_cases as dc    # noqa: E402


def make_case(seed):
    rng = np.random.default_rng(seed)
    nc = tuple(int(v) for v in rng.integers(3, 9, size=3))
    spacing = float(rng.choice([0.97, 0.98, 1.0, 1.05, 1.1]))
    jitter = float(rng.choice([0.005, 0.05, 0.15, 0.3])) if spacing >= 1.0 else 0.005
    poly = (0.7e-3, 1.0e-3) if rng.random() < 0.3 else None
    bed = synthetic.fcc_bed(nc, seed=seed, spacing=spacing, jitter=min(jitter, 0.45 * (spacing - 0.6)),
                            vmax=float(rng.choice([0.05, 0.2, 0.5])), poly=poly)
    periodic = bool(rng.random() < 0.7)
    walls = [(1, float(bed["boxlo"][1]), float(bed["boxhi"][1]))]
    if not periodic:
        bed["periodic"] = (0, 0, 0)
        bed["x"][:, 0] += 0.3e-3
        bed["x"][:, 2] += 0.3e-3
        bed["boxhi"][0] += 0.6e-3
        bed["boxhi"][2] += 0.6e-3
        walls += [(0, float(bed["boxlo"][0]), float(bed["boxhi"][0])), (2, float(bed["boxlo"][2]), float(bed["boxhi"][2]))]
    pair = str(rng.choice(["hertz", "hooke", "hooke_plain"]))
    cfg = dict(pair=pair, kn=1.0e7 if pair == "hertz" else 2.0e3, gamman=0.5 if pair == "hertz" else 50.0,
               xmu=float(rng.choice([0.0, 0.4, 0.8])), g=float(rng.choice([0.0, 9.81])), dt=1.0e-6,
               skin=float(rng.choice([0.04e-3, 0.08e-3, 0.25e-3])), dampflag=int(rng.random() < 0.8), walls=walls)
    if rng.random() < 0.3:
        y = bed["x"][:, 1]
        bed["type"] = np.where(y < np.quantile(y, 0.25), 2, 1).astype(np.int32)
        bed["v"][bed["type"] == 2] = 0.0
        cfg.update(frozen_types=[2], freeze_first=bool(rng.random() < 0.5), fdrag_group=str(rng.choice(["all", "active"])))
        if cfg["freeze_first"]:
            cfg["nve_all"] = bool(rng.random() < 0.5)
    # (fix cohesive after fix freeze is refused by the engine: no script of the reference has the two in that order)
    if pair == "hertz" and periodic and poly is None and rng.random() < 0.25 and not cfg.get("freeze_first"):
        cfg["cohesive"] = (1.0e-13, 1.0e-7, 1.0e-7, 1.0e-4, int(rng.integers(0, 2)))
    legs = (1, int(rng.integers(20, 70)), int(rng.integers(20, 70)))
    if pair == "hertz" and periodic and "frozen_types" not in cfg and rng.random() < 0.25:
        # pair lubricate/poly on top (hybrid/overlay): mu flaglog flagfld cutinner cutoff flagHI flagVF
        cfg["lub"] = (1.0e-3, int(rng.integers(0, 2)), int(rng.integers(0, 2)), 1.001 * float(np.max(bed["diameter"])),
                      1.1 * float(np.max(bed["diameter"])), 1, int(rng.integers(0, 2)))
    return bed, cfg, legs


def run_case(seed):
    bed, cfg, legs = make_case(seed)
    lmp = dc.make_hip(bed, cfg)
    orc = dc.make_oracle(bed, cfg)
    lmp.setup(); orc.setup()
    rng = np.random.default_rng(seed + 1)
    st = orc.get()
    fd = rng.normal(scale=1e-6, size=st["x"].shape)
    lmp.put_local_info(fd, st["tag"]); orc.put_fdrag(fd, st["tag"])
    d = float(np.max(bed["diameter"]))
    worst = 0.0
    for n in (0,) + legs:
        if n:
            lmp.step(n); orc.run(n)
        a, b = lmp.get_state(), orc.get()
        assert np.array_equal(a["tag"], b["tag"])
        ex = float(np.max(np.abs(a["x"] - b["x"])) / d)
        errs = [ex] + [dc.rel_err(a[k], b[k]) for k in ("v", "omega", "f")]
        # torques are sums of tangential forces alone: where those nearly vanish (frozen, frictionless or resting beds)
        # max |torque| is no scale -- the lever arm times the largest force is
        tscale = max(float(np.max(np.abs(b["torque"]))), 1e-3 * 0.5 * d * float(np.max(np.abs(b["f"]))))
        errs.append(float(np.max(np.abs(a["torque"] - b["torque"]))) / (tscale if tscale > 0 else 1.0))
        worst = max(worst, max(errs))
        # x, v, omega: the gates of SURVEY.md 8d.  Forces and torques get 1e-8: the sub-step in which a Hertz contact
        # forms, its damping term is already ~sqrt(overlap) large, and whether that is this sub-step or the next can hang
        # on the last bit of rsq (the GPU contracts dx*dx + dy*dy + dz*dz into FMAs, gcc on the host does not) -- seed
        # 1009: one grain's force off by 3e-10 of the largest for one sub-step, positions still equal to 1e-13 d
        assert max(errs[:3]) <= 1e-9 and max(errs[3:]) <= 1e-8, (n, errs)
        assert lmp.info().nbuilds == orc.nbuilds, (lmp.info().nbuilds, orc.nbuilds)
    ha, hb = lmp.history(), orc.history()
    assert set(ha) == set(hb), (len(ha), len(hb))
    return bed["n"], cfg["pair"], lmp.info().nbuilds, worst


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    bad = 0
    for seed in range(first, first + count):
        try:
            n, pair, nb, worst = run_case(seed)
            print("seed %d ok: n %d %s rebuilds %d worst %.1e" % (seed, n, pair, nb, worst), flush=True)
        except Exception as ex:   # noqa: BLE001
            bad += 1
            print("seed %d FAILED: %s" % (seed, str(ex)[:300]), flush=True)
            traceback.print_exc(limit=2)
    print("%d of %d cases failed" % (bad, count))
    sys.exit(1 if bad else 0)
