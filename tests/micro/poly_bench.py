"""development helper (numbers quoted in DESIGN.md): BASELINE config C4/C5's physics at 500 k grains on one GPU --
polydisperse bed, pair hybrid/overlay gran/hertzFix/history + lubricate/poly, fix cohesive -- sub-step kernel time
and particle-sub-steps/s"""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sedifoam_amd import synthetic, Lammps
n_target = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
mode = sys.argv[2] if len(sys.argv) > 2 else "all"   # hertz | cohesive | lub | all
bed = synthetic.fcc_bed(synthetic.fcc_cells_for(n_target), seed=5, poly=(0.85e-3, 1.0e-3), spacing=0.95)
lmp = Lammps()
lmp.set_box(bed["boxlo"], bed["boxhi"])
lmp.create_atoms(bed["x"], bed["diameter"], bed["density"], v=bed["v"])
for line in ["atom_style sphere", "boundary p f p", "newton off", "communicate single vel yes", "neighbor 0.2e-3 bin",
             "neigh_modify delay 0",
             ("pair_style hybrid/overlay gran/hertzFix/history 1e7 NULL 0.5 NULL 0.4 1 lubricate/poly 1.0e-3 1 1 1.001e-3 1.1e-3 1 1"
              if mode in ("lub", "all") else "pair_style gran/hertzFix/history 1e7 NULL 0.5 NULL 0.4 1"),
             "pair_coeff * *", "timestep 1e-6", "fix 1 all nve/sphere", "fix 2 all gravity 9.81 vector 0 -1 0",
             "fix 3 all fdrag",
             "fix ywall all wall/granFix 1e7 NULL 0.5 NULL 0.4 1 yplane %.17g %.17g" % (bed["boxlo"][1], bed["boxhi"][1]),
             ] + (["fix coh all cohesive 1.0e-13 1.0e-7 1.0e-7 1.0e-4 1"] if mode in ("cohesive", "all") else []):
    lmp.command(line)
lmp.setup()
info = lmp.info()
for _ in range(2):
    lmp.step(50)
lmp.set_profiling(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
for _ in range(steps):
    lmp.step(50)
torch.cuda.synchronize(); el = time.perf_counter() - t0
launches, ms = lmp.get_profile()
kh = info.npairs_full / 2.0 / info.nlocal
balg = 284.0 + 52.0 * kh   # SURVEY.md 8d, as bench.py
print(mode, "N %d  pairs/atom (full list) %.2f  kernel %.1f us  %.3e particle-substeps/s  rebuilds %d  frac(kernel) %.3f of 8 TB/s on %.0f B/particle-substep" % (
    info.nlocal, info.npairs_full / info.nlocal, 1e3 * ms / launches, info.nlocal * 50 * steps / el,
    lmp.info().nbuilds - info.nbuilds, balg * info.nlocal / (1e-3 * ms / launches) / 8e12, balg))
