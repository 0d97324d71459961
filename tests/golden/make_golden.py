#!/usr/bin/env python3
"""Collect the reference's own golden vectors for the hot path into tests/golden/.

Run in the build container only (needs /root/reference).  Copies DATA files (benchmark curves and
LAMMPS dump rows that the reference's auto-testing suite compares against by eye) -- no source.

  xiaocase3/data/lammps08.dat, xiaoCase3.dat : single 83 um sphere entrained by a 0.05 m/s flow
      (cases/auto-testing/test-cases/xiaocase3; columns: t vx vy vz / t vy)
  multiParticlesCollideRho/data/origin/p[1-4].dat : 4 spheres settling and colliding
      (rows = dump every 1000 DEM steps of 1e-5 s: id type d mass x y z vx vy vz)
"""
import os
import shutil

REF = "/root/reference/cases/auto-testing/test-cases"
HERE = os.path.dirname(os.path.abspath(__file__))
FILES = {
    "xiaocase3_lammps08.dat": "xiaocase3/data/lammps08.dat",
    "xiaocase3_xiaoCase3.dat": "xiaocase3/data/xiaoCase3.dat",
    "multiParticlesCollideRho_p1.dat": "multiParticlesCollideRho/data/origin/p1.dat",
    "multiParticlesCollideRho_p2.dat": "multiParticlesCollideRho/data/origin/p2.dat",
    "multiParticlesCollideRho_p3.dat": "multiParticlesCollideRho/data/origin/p3.dat",
    "multiParticlesCollideRho_p4.dat": "multiParticlesCollideRho/data/origin/p4.dat",
    "multiParticlesCollideDia_p1.dat": "multiParticlesCollideDia/data/origin/p1.dat",
    "multiParticlesCollideDia_p2.dat": "multiParticlesCollideDia/data/origin/p2.dat",
    "multiParticlesCollideDia_p3.dat": "multiParticlesCollideDia/data/origin/p3.dat",
    "multiParticlesCollideDia_p4.dat": "multiParticlesCollideDia/data/origin/p4.dat",
}
if __name__ == "__main__":
    for dst, src in FILES.items():
        shutil.copyfile(os.path.join(REF, src), os.path.join(HERE, dst))
        print("copied", src, "->", dst)
