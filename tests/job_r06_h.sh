#!/bin/bash
# full GPU suite without -x (every failure listed); C5_wide bench leg with and without the ghost-free build
cd $GRAFT_REPO_ROOT
(python -m pytest tests -m gpu -q 2>&1 | tail -60) > gpurun_out/r06_suite_h.log
for gf in 1 0; do
  echo "== SF_GHOST_FREE=$gf"
  SF_GHOST_FREE=$gf python - <<'P'
import sys, time
sys.path.insert(0, ".")
import numpy as np
import bench
from sedifoam_amd import synthetic
t0 = time.time()
bed = synthetic.grown_poly_bed(60000, seed=15, vmax=0.05, verbose=False)
print("grown in %.1f s, stages %d" % (time.time() - t0, bed["stages"]))
cfg = dict(kn=1e7, gamman=0.5, xmu=0.4, g=0.0, dt=1e-6, skin=0.06e-3, walls=[], cohesive=bench.C5W_COHESIVE, lub=bench.C5W_LUB)
from sedifoam_amd.lammps import Lammps
from tests import dem_cases as dc
lmp = dc.make_hip(bed, dict(cfg, pair="hertz"))
lmp.setup()
try:
    for k in range(8):
        lmp.step(50)
        st = lmp.get_state()
        print(k, "nbuilds", lmp.info().nbuilds, "max|v|", float(np.abs(st["v"]).max()), "finite", bool(np.isfinite(st["x"]).all()))
except Exception as ex:
    print("ERROR", ex)
P
done > gpurun_out/r06_c5w_repro.txt 2>&1
tail -25 gpurun_out/r06_suite_h.log; cat gpurun_out/r06_c5w_repro.txt
