"""development helper: the gran/hooke/history law (what the reference's cases run) on the 1M bed"""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from sedifoam_amd import synthetic
import bench
bed = synthetic.fcc_bed(synthetic.fcc_cells_for(1000000), seed=12348)
script = synthetic.hertz_script(bed, kn=2.0e3, gamman=50.0, xmu=0.4, dt=1.0e-6, skin_d=0.25, g=9.81,
                                pair="gran/hooke/history", wall="wall/gran")
lmp = bench.build_engine(bed, script); lmp.setup()
for _ in range(2):
    lmp.step(50)
lmp.set_profiling(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8):
    lmp.step(50)
torch.cuda.synchronize(); el = time.perf_counter() - t0
launches, ms = lmp.get_profile()
print("hooke 1M: kernel %.1f us  %.3e particle-substeps/s" % (1e3 * ms / launches, bed["n"] * 400 / el))
