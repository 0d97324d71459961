"""Seeded differential campaign of the coupled step (drag closure + force assembly, DEM sub-steps, cell owner, scatter,
diffusion smoothing, calcTcFields), HIP against the oracle: random force switches, drag model, sub-cycling, mesh and
smoothing parameters through tests/test_cloud_gpu.py::_coupled_case (development helper, GPU box).
usage: python tests/fuzz_cloud.py [first_seed] [cases]"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sedifoam_amd import synthetic        # noqa: E402
from tests import test_cloud_gpu as tc    # noqa: E402


def run_case(seed):
    rng = np.random.default_rng(seed)
    flags = {k: bool(rng.random() < p) for k, p in (("particleDrag", 0.85), ("particlePressureGrad", 0.7),
                                                    ("particleBuoyancy", 0.3), ("particleAddedMass", 0.4),
                                                    ("particleLift", 0.4), ("lubricationForce", 0.0),
                                                    ("particleHistoryForce", 0.25))}
    drag = str(rng.choice(["ErgunWenYu", "SyamlalOBrien"]))
    nc = tuple(int(v) for v in rng.integers(5, 9, size=3))
    bed = synthetic.fcc_bed(nc, seed=seed, vmax=float(rng.choice([0.02, 0.1])), spacing=1.06)   # (solid fraction 0.62)
    # cells at least ~2.9 d wide (centre-counted alpha stays below 0.85)
    ext = (bed["boxhi"] - bed["boxlo"]) / 1.0e-3
    mesh_n = tuple(int(max(1, min(int(e / float(os.environ.get("FUZZ_MESH_DIV", "2.9"))), int(rng.integers(2, 7))))) for e in ext)
    smooth = None
    if rng.random() < 0.5:
        smooth = dict(diffusionBandWidth=float(rng.choice([0.003, 0.006])), diffusionSteps=int(rng.integers(2, 7)))
        for k in ("UfSmooth", "UpSmooth", "dragSmooth", "alphaSmooth"):
            smooth[k] = int(rng.random() < 0.8)
    extra = {}
    if flags["particleAddedMass"] and rng.random() < 0.5:
        extra["carrier_rho"] = 1000.0
    if os.environ.get("FUZZ_VERBOSE"):
        print("  case", seed, drag, flags, nc, mesh_n, smooth, extra, flush=True)
    # added mass without drag: the assembled force is (DuDt - (v - vOld) / deltaT) and little else -- the difference of
    # two nearly equal velocities over 50 us, against which a 1e-16 of v is 1e-11 (seeds 108, 119 with 2.4 d cells)
    tol = 1e-10 if flags["particleAddedMass"] and not flags["particleDrag"] else None
    tc._coupled_case(drag, flags, sub_cycles=int(rng.integers(1, 4)), n_cfd=int(rng.integers(2, 4)), smooth=smooth,
                     deltaT=float(rng.choice([50e-6, 100e-6])), bed=bed, mesh_n=mesh_n, cfg_extra=extra, tol=tol)
    return bed["n"], drag, mesh_n, {k for k, v in flags.items() if v}, bool(smooth)


_last = []
_rel = tc.dc.rel_err


def _rel_logged(a, b):
    v = _rel(a, b)
    _last.append(v)
    del _last[:-6]
    return v


tc.dc.rel_err = _rel_logged


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    bad = 0
    for seed in range(first, first + count):
        try:
            print("seed %d ok:" % seed, *run_case(seed), flush=True)
        except Exception as ex:   # noqa: BLE001
            bad += 1
            print("seed %d FAILED: %s  last rel_err values %s" % (seed, str(ex)[:400], ["%.2e" % v for v in _last]), flush=True)
            traceback.print_exc(limit=3)
    print("%d of %d cases failed" % (bad, count))
    sys.exit(1 if bad else 0)
