"""development helper: wall time of the phases of SlabDriver.rebuild() (decomposed path, RCCL self images), 1M atoms"""
import os, sys, time
os.environ.setdefault("SF_HALO_SELF_COMM", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29542")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
sys.path.insert(0, "/root/repo")
import numpy as np, torch, torch.distributed as dist
from sedifoam_amd import synthetic
from sedifoam_amd.halo import SlabDriver, BORDER_DOUBLES
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
kw = dict(kn=1.0e7, gamman=0.5, xmu=0.4, dt=1.0e-6, skin_d=0.25, g=9.81)
bed = synthetic.fcc_bed(synthetic.fcc_cells_for(1000000), seed=12348)
self = SlabDriver.from_bed(bed, synthetic.hertz_script(bed, **kw), dist, 0, 1)
self.setup(); self.step(50)
acc = {}
def tick(name, t0):
    torch.cuda.synchronize(); t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t - t0); return t
R = 6
for rep in range(R):
    e = self.e
    torch.cuda.synchronize(); t = time.perf_counter()
    e.rebuild_begin(); t = tick("begin", t)
    crossed = e.migrate_count(); v = self._allreduce_max(int(e.info().max_neigh_used) + ((1 << 20) if crossed else 0)); e.migrate_set_slots(v & ((1 << 20) - 1)); t = tick("count+allreduce+slots", t)
    rec = e.migrate_record_doubles()
    nmax = max(self._cap_atoms // 8, 1024)
    b0 = self._buf("mig_l", nmax * rec); b1 = self._buf("mig_r", nmax * rec)
    n0 = e.migrate_pack(0, self.shift_left, b0); n1 = e.migrate_pack(1, self.shift_right, b1); t = tick("migrate_pack", t)
    rl, ml, rr, mr = self._exchange(b0, n0, b1, n1); t = tick("migrate_exchange", t)
    e.migrate_unpack(rl, ml); e.migrate_unpack(rr, mr); t = tick("migrate_unpack", t)
    e.rebuild_sort(); t = tick("sort", t)
    cap = max(self._cap_atoms, e.info().nlocal)
    s0 = self._buf("bor_l", cap * BORDER_DOUBLES); s1 = self._buf("bor_r", cap * BORDER_DOUBLES)
    a0 = e.border_pack(0, self.shift_left, s0); a1 = e.border_pack(1, self.shift_right, s1); t = tick("border_pack", t)
    self._nsend = [a0, a1]
    rl, ml, rr, mr = self._exchange(s0, a0 * BORDER_DOUBLES, s1, a1 * BORDER_DOUBLES); t = tick("border_exchange", t)
    self._nrecv = [ml // BORDER_DOUBLES, mr // BORDER_DOUBLES]
    e.border_unpack(0, rl, self._nrecv[0]); e.border_unpack(1, rr, self._nrecv[1]); t = tick("border_unpack", t)
    e.rebuild_finish(); t = tick("finish", t)
for k, v in acc.items():
    print("%-18s %.3f ms" % (k, 1e3 * v / R))
print("sum %.3f ms (every phase followed by a device sync)" % (1e3 * sum(acc.values()) / R))
dist.destroy_process_group()
