"""development helper: soak run of the final code -- 6000 sub-steps of the 1M bench bed (the bed expands: ~60 rebuilds),
then 60 coupled steps with smoothing on the same engine; prints energy, extrema, rebuild count, finiteness"""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sedifoam_amd import synthetic, enhancedCloud
import bench
kw = dict(kn=1.0e7, gamman=0.5, xmu=0.4, dt=1.0e-6, skin_d=0.25, g=9.81)
bed = synthetic.fcc_bed(synthetic.fcc_cells_for(1000000), seed=12349)
lmp = bench.build_engine(bed, synthetic.hertz_script(bed, **kw)); lmp.setup()
m = np.pi / 6 * bed["diameter"] ** 3 * bed["density"]
t0 = time.perf_counter()
for k in range(12):
    lmp.step(500)
    st = lmp.get_local_info()
    v = st["v"]; o = np.argsort(st["tag"])
    ke = 0.5 * (m * (v[o] ** 2).sum(axis=1)).sum()
    print("substeps %5d  KE %.4e  max|v| %.4f  y [%.6f, %.6f]  finite %s  builds %d  max_neigh %d" % (
        500 * (k + 1), ke, np.abs(v).max(), st["x"][:, 1].min(), st["x"][:, 1].max(),
        bool(np.isfinite(st["x"]).all() and np.isfinite(v).all()), lmp.info().nbuilds, lmp.info().max_neigh_used))
print("DEM wall %.2f s" % (time.perf_counter() - t0))
lo = np.array(bed["boxlo"]); hi = np.array(bed["boxhi"])
mesh_n = np.maximum(1, np.round((hi - lo) / 3.0e-3)).astype(np.int32)
cloud = enhancedCloud(lmp, lo, (hi - lo) / mesh_n, mesh_n,
                      dict(dragModel="ErgunWenYu", subCycles=1, g=(0, -9.81, 0), diffusionBandWidth=6e-3, diffusionSteps=6,
                           maxPossibleAlpha=0.65), dict(rhob=1000.0, nub=1e-6), 50e-6)
nc = int(mesh_n.prod())
cloud.setFluid(Uf=np.tile([0.0, 0.05, 0.0], (nc, 1)), gradp=np.tile([0.0, -9810.0, 0.0], (nc, 1)))
t0 = time.perf_counter()
for k in range(60):
    cloud.evolve(); cloud.calcTcFields()
    if k % 20 == 19:
        g = cloud.gamma(); A = cloud.Asrc()
        print("coupled %3d  gamma [%.4f, %.4f]  |Asrc|max %.4e  finite %s  builds %d" % (
            k + 1, g.min(), g.max(), np.abs(A).max(), bool(np.isfinite(g).all() and np.isfinite(A).all()),
            lmp.info().nbuilds))
print("coupled wall %.2f s" % (time.perf_counter() - t0))
