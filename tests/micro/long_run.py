"""development helper: 5000 sub-steps of the 1M bench bed; prints kinetic energy, max speed, rebuilds, finiteness"""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from sedifoam_amd import synthetic
import bench
kw = dict(kn=1.0e7, gamman=0.5, xmu=0.4, dt=1.0e-6, skin_d=0.25, g=9.81)
bed = synthetic.fcc_bed(synthetic.fcc_cells_for(1000000), seed=12348)
lmp = bench.build_engine(bed, synthetic.hertz_script(bed, **kw)); lmp.setup()
m = np.pi / 6 * bed["diameter"] ** 3 * bed["density"]
t0 = time.perf_counter()
for k in range(10):
    lmp.step(500)
    st = lmp.get_local_info()
    v = st["v"]
    ke = 0.5 * (m[st["tag"] - 1] * (v ** 2).sum(axis=1)).sum()
    print("substeps %5d  KE %.4e  max|v| %.4f  ymin %.6f ymax %.6f  finite %s  builds %d" % (
        (k + 1) * 500, ke, np.abs(v).max(), st["x"][:, 1].min(), st["x"][:, 1].max(), np.isfinite(st["x"]).all() and np.isfinite(v).all(), lmp.info().nbuilds))
print("wall %.2f s" % (time.perf_counter() - t0))
