"""development helper (GPU box): the walled-x 2 x 1 x 2 brick case of tests/test_halo_gpu.py under a transport, error
against the single-domain run after growing numbers of sub-steps.  usage: SF_HALO_DIRECT=2 python tests/micro/debug_gs_walled.py"""
import os, sys, socket, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch.multiprocessing as mp
from tests import dem_cases as dc
from tests import test_halo_gpu as H
from tests import test_dem_gpu as T

if __name__ == "__main__":
    os.environ.setdefault("SF_HALO_DIRECT_TIMEOUT", "20")
    grid, ncells = (2, 1, 2), (8, 5, 8)
    periodic_x = len(sys.argv) > 1 and sys.argv[1] == "periodic"
    tmp = tempfile.mkdtemp()
    import pathlib
    lib = H._standin_rccl(pathlib.Path(tmp))
    for steps in ((1,), (2,), (5,), (10,), (20,), (50,), (50, 50)):
        bed = T._bed(ncells, periodic=True, seed=41, vmax=0.5)
        cfg = dict(T.BASE, skin=0.05e-3)
        cfg["walls"] = T._walls(bed)
        if not periodic_x:
            bed, cfg = H._walled_x(bed, cfg)
        ref = dc.make_hip(bed, cfg)
        ref.setup()
        for n in steps:
            ref.step(n)
        a = ref.get_state()
        nb = ref.info().nbuilds
        del ref
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        out = tempfile.mkdtemp()
        mp.spawn(H._two_rank_worker, args=(4, port, out, steps, False, "hertz", "rccl", lib, ncells, periodic_x, grid), nprocs=4, join=True)
        parts = [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(4)]
        tag = np.concatenate([p["tag"] for p in parts]); order = np.argsort(tag); oa = np.argsort(a["tag"])
        line = "steps %-10s builds(ref) %d rebuilds %s direct %s :" % (steps, nb, [int(p["rebuilds"]) for p in parts], [int(p["direct"]) for p in parts])
        for k in ("x", "v", "f"):
            got = np.concatenate([p[k] for p in parts])[order]; want = a[k][oa]
            if k == "x":
                for d in range(3):
                    if bed["periodic"][d]:
                        Ld = bed["boxhi"][d] - bed["boxlo"][d]
                        got[:, d] = np.mod(got[:, d] - bed["boxlo"][d], Ld); want = want.copy(); want[:, d] = np.mod(want[:, d] - bed["boxlo"][d], Ld)
            err = np.abs(got - want).max(axis=1)
            worst = int(np.argmax(err))
            line += "  %s max %.2e (tag %d at x=%s)" % (k, err.max(), int(tag[order][worst]), np.round(a["x"][oa][worst] * 1e3, 3))
        print(line, flush=True)
