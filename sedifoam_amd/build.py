"""Build libsedifoam_amd.so in-tree with hipcc for gfx950 (no GPU needed: hipcc cross-compiles)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libsedifoam_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-file flags.  sf_dem.hip (the sub-step kernel): the ILP-first machine scheduler orders the loop's loads and the FP64
# chains so that the kernel runs 1.7-1.9 % faster at 1 M grains on three boxes (2.7 % at 2 M, neutral below 130 k; same
# registers, no scratch; results bit-identical) -- profiles/r04_README.md section 3
FILE_FLAGS = {"sf_dem.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "sedifoam_amd.h"))
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def kernel_source_hash():
    """sha256 (first 16 hex digits) of the sources the sub-step kernel is compiled from -- their CODE: comments and white
    space are taken out first, so that a corrected comment does not disown a measurement -- stored next to a PMC summary
    (tests/pmc_summarize.py) so that bench.py can tell whether the committed HBM-traffic counters still describe the
    kernel it is timing"""
    import hashlib
    import re
    h = hashlib.sha256()
    for f in ("sf_dem_kernels.h", "sf_dem_variants.h", "sf_dem_gs.h", "sf_physics.h", "sf_dem.h", "sf_common.h"):
        src = open(os.path.join(CSRC, f), "r").read()
        if f == "sf_dem.h":
            # (the structures the kernel is compiled against -- DemPtrs, StepParams, BinGrid, the flag words, the list-word
            # bits -- stand in front of the host-side declarations, which are not kernel source: a new member of the engine
            # class or of the rebuild predictor does not disown a measurement)
            cut = src.find("struct RebuildPredictor")
            if cut > 0:
                src = src[:cut]
        src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)     # (none of these files holds "//" or "/*" inside a string)
        src = re.sub(r"//[^\n]*", " ", src)
        h.update(" ".join(src.split()).encode())
    h.update(" ".join(FLAGS + FILE_FLAGS.get("sf_dem.hip", [])).encode())   # (and the flags it is compiled with)
    return h.hexdigest()[:16]


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = _headers()
    jobs = []
    objs = []
    for src in _sources():
        obj = os.path.join(OBJDIR, src[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [os.path.join(CSRC, src), os.path.abspath(__file__)] + hdrs):   # (the flags live in this file)
            jobs.append([HIPCC] + FLAGS + FILE_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        return r

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
