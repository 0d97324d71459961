"""Random processor grids and beds through tests/test_halo_gpu.py::test_cxx_brick_driver_on_a_processor_grid (C++ brick
driver over the stand-in wire against the single-domain run); development helper, GPU box.
Both transports: the wire (SF_HALO_DIRECT=0) and direct ghost writes (=1, grids of up to 6 ranks: more spinning processes
than that time-slice the one GPU for minutes).
usage: [FUZZ_DIRECT=2] python tests/fuzz_bricks.py [seed] [cases]   (42 cases on the final code of round 3, 40 on that of
round 4, 80 (seeds 7 and 11) with FUZZ_DIRECT=2 -- ghost slots -- on that of round 5: none failing)"""
import os, sys, tempfile, pathlib, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_halo_gpu as th


class _Env:
    """pytest's monkeypatch.setenv, for calling the test function directly"""

    def setenv(self, k, v):
        os.environ[k] = v


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    grids = [(2, 1, 1), (3, 1, 1), (2, 1, 2), (1, 1, 2), (1, 1, 3), (2, 2, 1), (1, 2, 1), (2, 2, 2), (3, 1, 2), (4, 1, 1), (2, 1, 3)]
    bad = 0
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
        g = grids[int(rng.integers(len(grids)))]
        # every brick at least 3 lattice cells (4 d) wide along a cut dimension
        nc = tuple(int(g[d] * rng.integers(3, 6)) if g[d] > 1 else int(rng.integers(4, 8)) for d in range(3))
        physics = str(rng.choice(["hertz", "hertz", "loose", "c5"]))
        if physics == "c5" and g[1] > 1:
            physics = "hertz"
        px = bool(rng.random() < 0.8) or physics != "hertz"
        direct = "1" if (g[0] * g[1] * g[2] <= 6 and rng.random() < 0.5) else "0"
        if os.environ.get("FUZZ_DIRECT") and g[0] * g[1] * g[2] <= 6:   # (FUZZ_DIRECT=2: every case of up to 6 ranks on ghost slots)
            direct = os.environ["FUZZ_DIRECT"]
        c = (g, nc, physics, px, direct)
        with tempfile.TemporaryDirectory() as d:
            try:
                th.test_cxx_brick_driver_on_a_processor_grid(pathlib.Path(d), _Env(), *c)
                print("ok", c, flush=True)
            except Exception as ex:
                bad += 1
                print("FAILED", c, str(ex)[:300], flush=True)
                traceback.print_exc(limit=4)
    print(bad, "failed")


if __name__ == "__main__":
    main()
