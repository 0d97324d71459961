#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_dem_gpu.py -x -q -k "variants_agree" 2>&1 | tail -3
SF_PARK_MARGIN=-6 SF_DEBUG_HIST=1 python - <<'P' 2>&1 | tail -3
import sys
sys.path.insert(0, ".")
import numpy as np
import tests.test_dem_gpu as t
from tests import dem_cases as dc
bed = t._bed((6, 6, 6), periodic=True, seed=99, vmax=0.5)
lmp = dc.make_hip(bed, dict(t.BASE, skin=0.05e-3, walls=t._walls(bed)))
lmp.setup(); lmp.step(120)
print("nbuilds", lmp.info().nbuilds, "max_neigh_used", lmp.info().max_neigh_used)
P
