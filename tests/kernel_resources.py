"""Register / scratch / occupancy table of the sub-step kernels (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tests/kernel_resources.py [-DFLAG ...]     (development helper, CPU only: hipcc cross-compiles)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
       "-mllvm", "-amdgpu-sched-strategy=max-ilp",   # (sedifoam_amd/build.py FILE_FLAGS for this file)
       "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(ROOT, "sedifoam_amd", "csrc", "sf_dem.hip"),
       "-o", "/tmp/_kr.o"] + sys.argv[1:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\S+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for name, r in rows.items():
    if "k_substep" not in name and "--all" not in sys.argv:
        continue
    short = re.sub(r"\(.*", "", name).replace("void sf::", "")
    print("%-46s VGPR %4s  scratch %4s B  waves/SIMD %s" % (short, r.get("VGPRs"), r.get("ScratchSize [bytes/lane]"),
                                                          r.get("Occupancy [waves/SIMD]")))
