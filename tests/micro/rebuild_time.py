"""development helper: wall time of SlabDriver.rebuild() (decomposed path, RCCL self images) vs the single-domain
engine's rebuild, 1M atoms"""
import os, sys, time
os.environ.setdefault("SF_HALO_SELF_COMM", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
sys.path.insert(0, "/root/repo")
import numpy as np, torch, torch.distributed as dist
from sedifoam_amd import synthetic
from sedifoam_amd.halo import SlabDriver
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
kw = dict(kn=1.0e7, gamman=0.5, xmu=0.4, dt=1.0e-6, skin_d=0.25, g=9.81)
bed = synthetic.fcc_bed(synthetic.fcc_cells_for(1000000), seed=12348)
drv = SlabDriver.from_bed(bed, synthetic.hertz_script(bed, **kw), dist, 0, 1)
drv.setup()
drv.step(50)
import cProfile, pstats
ts = []
for k in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    drv.rebuild()
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("decomposed rebuild ms:", ["%.2f" % (1e3 * t) for t in ts])
pr = cProfile.Profile(); pr.enable(); drv.rebuild(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
dist.destroy_process_group()
