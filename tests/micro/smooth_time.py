import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch
from sedifoam_amd import synthetic, enhancedCloud, Lammps
import bench
kw = dict(kn=1.0e7, gamman=0.5, xmu=0.4, dt=1.0e-6, skin_d=0.25, g=9.81)
bed = synthetic.fcc_bed(synthetic.fcc_cells_for(1000000), seed=12348)
script = synthetic.hertz_script(bed, **kw)
lmp = bench.build_engine(bed, script); lmp.setup()
mesh_n = np.clip(((bed["boxhi"] - bed["boxlo"]) / 3.0e-3).astype(int), 1, 32)
dx = (bed["boxhi"] - bed["boxlo"]) / mesh_n
for band in (0.0, 6e-3):
    cloud = enhancedCloud(lmp, bed["boxlo"], dx, mesh_n, dict(dragModel="ErgunWenYu", subCycles=1, maxPossibleAlpha=0.65, diffusionBandWidth=band, diffusionSteps=6),
                          dict(rhob=1000.0, nub=1.0e-6), deltaT=50e-6)
    nc = int(np.prod(mesh_n))
    cloud.setFluid(Uf=np.tile([0.0, 0.05, 0.0], (nc, 1)), gradp=np.tile([0.0, -9810.0, 0.0], (nc, 1)))
    cloud.calcTcFields()
    cloud.evolve(); cloud.calcTcFields()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(4):
        cloud.evolve(); cloud.calcTcFields()
    torch.cuda.synchronize(); print("band", band, "coupled step ms", (time.perf_counter()-t)/4*1e3, "mesh", mesh_n)
    f = np.random.default_rng(0).uniform(size=(nc,3))
    t=time.perf_counter(); g = cloud.smoothField(f); print("  smoothField(vector) ms", (time.perf_counter()-t)*1e3)
    cloud.close()
