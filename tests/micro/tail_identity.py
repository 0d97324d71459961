"""development helper (GPU): the two-lane launch tail (SF_TAIL_FRAC) must leave every bit of the state as the plain
one-lane launch leaves it.  usage: python tests/micro/tail_identity.py [N] [frac] [pos] [jitter]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sedifoam_amd import synthetic, Lammps

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
frac = sys.argv[2] if len(sys.argv) > 2 else "0.1"
pos = sys.argv[3] if len(sys.argv) > 3 else "0"
jit = float(sys.argv[4]) if len(sys.argv) > 4 else 0.005


def run(env):
    for k in ("SF_TAIL_FRAC", "SF_TAIL_POS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    os.environ["SF_LPA"] = "1"
    bed = synthetic.fcc_bed(synthetic.fcc_cells_for(n), seed=77, jitter=jit, spacing=0.98 if jit < 0.1 else 1.1)
    lmp = Lammps()
    lmp.set_box(bed["boxlo"], bed["boxhi"])
    lmp.create_atoms(bed["x"], bed["diameter"], bed["density"], v=bed["v"])
    for line in synthetic.hertz_script(bed):
        lmp.command(line)
    lmp.setup()
    lmp.step(120)
    st = lmp.get_state()
    return st, lmp.info().nbuilds


a, ba = run({})
b, bb = run({"SF_TAIL_FRAC": frac, "SF_TAIL_POS": pos})
ok = ba == bb
for k in ("tag", "x", "v", "omega", "f"):
    same = np.array_equal(a[k], b[k])
    ok = ok and same
    print(k, "identical" if same else "DIFFERENT max|d| %.3e" % float(np.max(np.abs(a[k] - b[k]))))
print("rebuilds", ba, bb, "=> BIT-IDENTICAL" if ok else "=> MISMATCH")
sys.exit(0 if ok else 1)
