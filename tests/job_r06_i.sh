#!/bin/bash
# list build with parked candidates in LDS + old list read in place: parity tests, A/B of the rebuild; what blows up in C5_wide
cd $GRAFT_REPO_ROOT
(python -m pytest tests/test_dem_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py tests/test_cloud_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -15) > gpurun_out/r06_suite_i.log
{
for rep in 1 2; do
tests/ab_env.sh "--bed fluidised --no-fluidised --no-parity" "SF_HIST_IN_PLACE=0 SF_BUILD_LDS=0" "SF_HIST_IN_PLACE=0 SF_BUILD_LDS=1" "SF_HIST_IN_PLACE=1"
SF_LIB_PATH=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_bw4.so tests/ab_env.sh "--bed fluidised --no-fluidised --no-parity" "SF_HIST_IN_PLACE=1"
tests/ab_env.sh "--bed fluidised --particles 100000 --no-fluidised --no-parity" "SF_HIST_IN_PLACE=0 SF_BUILD_LDS=0" "SF_HIST_IN_PLACE=0 SF_BUILD_LDS=1" "SF_HIST_IN_PLACE=1"
SF_LIB_PATH=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_bw4.so tests/ab_env.sh "--bed fluidised --particles 100000 --no-fluidised --no-parity" "SF_HIST_IN_PLACE=1"
done
tests/ab_env.sh "--no-fluidised --no-parity" "SF_HIST_IN_PLACE=0 SF_BUILD_LDS=0" "SF_HIST_IN_PLACE=1"
} > gpurun_out/r06_build_lds_ab.txt 2>&1
python - > gpurun_out/r06_c5w_terms.txt 2>&1 <<'P'
import sys, time
sys.path.insert(0, ".")
import numpy as np
import bench
from sedifoam_amd import synthetic
from tests import dem_cases as dc
bed = synthetic.grown_poly_bed(8000, seed=15, vmax=0.05)
base = dict(kn=1e7, gamman=0.5, xmu=0.4, g=0.0, dt=1e-6, skin=0.06e-3, walls=[], pair="hertz")
cases = {"hertz only": {}, "+cohesive": dict(cohesive=bench.C5W_COHESIVE), "+lub": dict(lub=bench.C5W_LUB),
         "+lub flaglog 0": dict(lub=(1.0e-3, 0, 0, 1.001 * 1.5e-3, 1.1 * 1.5e-3, 1, 1)),
         "+lub narrow-style cutoffs 1.001e-3/1.1e-3": dict(lub=(1.0e-3, 1, 0, 1.001e-3, 1.1e-3, 1, 1))}
for name, extra in cases.items():
    cfg = dict(base, **extra)
    lmp = dc.make_hip(bed, cfg); lmp.setup()
    orc = dc.make_oracle(bed, cfg); orc.setup()
    out = []
    try:
        for k in range(4):
            lmp.step(50); orc.run(50)
            a, b = lmp.get_state(), orc.get()
            out.append("%.3g/%.3g" % (float(np.abs(a["v"]).max()), float(np.abs(b["v"]).max())))
    except Exception as ex:
        out.append("ERROR " + str(ex)[:60])
    print("%-45s max|v| hip/oracle after 50,100,150,200: %s" % (name, "  ".join(out)))
P
tail -3 gpurun_out/r06_suite_i.log; cat gpurun_out/r06_build_lds_ab.txt gpurun_out/r06_c5w_terms.txt
