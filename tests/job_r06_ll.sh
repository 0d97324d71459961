#!/bin/bash
# pair-split record loads in the quad list build (experiment arms 14 / 18) against 4 lanes: bit identity, build kernel by trace
cd $GRAFT_REPO_ROOT
python - <<'P'
import os, sys
sys.path.insert(0, ".")
import numpy as np
import tests.test_dem_gpu as t
from tests import dem_cases as dc
bed = t._bed((7, 6, 8), periodic=True, seed=23, vmax=0.5, jitter=0.3, spacing=1.1)
outs = []
for q in ("4", "14", "18"):
    os.environ["SF_BUILD_QUAD"] = q
    lmp = dc.make_hip(bed, dict(t.BASE, skin=0.03e-3, walls=t._walls(bed)))
    lmp.setup(); lmp.step(150)
    outs.append((lmp.get_state(), lmp.info().nbuilds))
for o, nb in outs[1:]:
    print("nbuilds", nb, outs[0][1], "bitwise equal:", all(np.array_equal(o[k], outs[0][0][k]) for k in ("x", "v", "omega", "f", "torque")))
P
for q in 4 14 18; do
SF_BUILD_QUAD=$q SF_TRACE_MIN_NS=12000 tests/trace_rebuild.sh r06_c3ll$q "--bed fluidised --particles 100000 --no-fluidised --no-parity" > gpurun_out/r06_trace_c3ll$q.txt 2>&1
SF_BUILD_QUAD=$q tests/trace_rebuild.sh r06_l1mll$q "--bed fluidised --no-fluidised --no-parity" > gpurun_out/r06_trace_l1mll$q.txt 2>&1
echo "SF_BUILD_QUAD=$q: $(grep -h 'k_build_neigh' gpurun_out/r06_trace_c3ll$q.txt gpurun_out/r06_trace_l1mll$q.txt | awk '{print $4}' | tr '\n' ' ')"
done
rm -rf gpurun_out/kt_r06_*ll*
