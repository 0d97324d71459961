"""world_size-2 (and -3) CPU test of the multi-rank path: the product's SlabDriver protocol
(sedifoam_amd/halo.py: migration with shear history, border ghosts, per-sub-step forward halo, rebuild vote)
over torch.distributed/gloo, with the CPU oracle standing in for the HIP engine behind the same adaptor
interface and the same record layouts.  The decomposed run must reproduce the single-domain oracle."""
import os
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """(kept name) a FileStore rendezvous, not a port: see tests/rdzv.py"""
    sys.path.insert(0, ROOT)
    from tests.rdzv import new_rendezvous
    return new_rendezvous()


def _case(periodic_x=True, vmax=0.5, skin=0.05e-3, seed=31, physics="hertz"):
    sys.path.insert(0, ROOT)
    from sedifoam_amd import synthetic
    if physics == "c5":
        # BASELINE config C5's physics: polydisperse grains, fix cohesive + hybrid/overlay lubricate/poly with the
        # volume-fraction (flagVF 1) and isotropic FLD terms (flagfld 1) that need the particle volume of ALL ranks
        bed = synthetic.fcc_bed((8, 4, 4), seed=seed, vmax=vmax, poly=(0.85e-3, 1.0e-3), spacing=0.95)
        cfg = dict(pair="hertz", kn=1.0e7, gamman=0.5, xmu=0.4, g=9.81, dt=1.0e-6, skin=skin,
                   cohesive=(1.0e-13, 1.0e-7, 1.0e-7, 1.0e-4, 1), lub=(1.0e-3, 1, 1, 1.001e-3, 1.1e-3, 1, 1),
                   walls=[(1, float(bed["boxlo"][1]), float(bed["boxhi"][1]))])
        return bed, cfg
    bed = synthetic.fcc_bed((8, 4, 4), seed=seed, vmax=vmax)
    if not periodic_x:
        bed["periodic"] = (0, 0, 1)
        bed["x"][:, 0] += 0.3e-3
        bed["boxhi"][0] += 0.6e-3
    cfg = dict(pair="hertz", kn=1.0e7, gamman=0.5, xmu=0.4, g=9.81, dt=1.0e-6, skin=skin,
               walls=[(1, float(bed["boxlo"][1]), float(bed["boxhi"][1]))])
    if not periodic_x:
        cfg["walls"].append((0, float(bed["boxlo"][0]), float(bed["boxhi"][0])))
    return bed, cfg


def _worker(rank, world, port, outdir, periodic_x, steps, mode, physics="hertz"):
    sys.path.insert(0, ROOT)
    os.environ["SF_HALO_FUSED"] = "0" if mode == "p2p+allreduce" else "1"
    os.environ["SF_HALO_OVERLAP"] = "1" if mode == "overlap" else "0"
    import torch.distributed as dist
    from oracle import binding as ob
    from sedifoam_amd.halo import SlabDriver
    from tests import dem_cases as dc
    dist.init_process_group("gloo", init_method=port, rank=rank, world_size=world)
    bed, cfg = _case(periodic_x, physics=physics)
    lo, hi = float(bed["boxlo"][0]), float(bed["boxhi"][0])
    dem = dc.make_oracle(dc.subset(bed, dc.slab_mask(bed, rank, world)), cfg)
    drv = SlabDriver(ob.OracleSlabEngine(dem), dist, rank, world, lo, hi, periodic_x=periodic_x,
                     transport="host" if mode == "fused-p2p" else "direct")
    assert drv.fused == (mode != "p2p+allreduce") and drv.overlap == (mode == "overlap")
    drv.setup()
    for n in steps:
        drv.step(n)
    st = dem.get()
    h = dem.history()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), x=st["x"], v=st["v"], omega=st["omega"], f=st["f"],
             torque=st["torque"], tag=st["tag"], rebuilds=drv.n_rebuilds,
             hk=np.array(sorted(h), dtype=np.int64).reshape(-1, 2),
             hv=np.array([h[k] for k in sorted(h)]).reshape(-1, 3))
    dist.barrier()
    dist.destroy_process_group()


# mode: "fused" = one all_to_all_single per sub-step (halo + rebuild vote; what RCCL runs), "fused-p2p" = the same
# chunks as point-to-point messages (transport="host"), "p2p+allreduce" = the older two-collective protocol,
# "overlap" = the boundary / exchange / interior schedule of the overlapped halo (control flow only on the CPU twin)
@pytest.mark.parametrize("world,periodic_x,mode", [(2, True, "fused"), (3, True, "fused"), (2, False, "fused"),
                                                   (3, False, "fused-p2p"), (3, True, "p2p+allreduce"),
                                                   (3, True, "overlap"), (2, False, "overlap")])
def test_decomposed_run_matches_single_domain(world, periodic_x, mode):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from tests import dem_cases as dc
    steps = (40, 40)
    bed, cfg = _case(periodic_x)
    ref = dc.make_oracle(bed, cfg)
    ref.setup()
    for n in steps:
        ref.run(n)
    a = ref.get()
    ha = ref.history()
    assert ref.nbuilds >= 3      # the run crosses several rebuilds (migration + history carry-over)
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_worker, args=(world, _free_port(), out, periodic_x, steps, mode), nprocs=world, join=True)
        parts = [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(world)]
    tag = np.concatenate([p["tag"] for p in parts])
    assert len(tag) == bed["n"] and len(np.unique(tag)) == bed["n"]      # nobody lost or duplicated
    order = np.argsort(tag)
    for k in ("x", "v", "omega", "f", "torque"):
        got = np.concatenate([p[k] for p in parts])[order]
        if k == "x" and periodic_x:
            L = bed["boxhi"][0] - bed["boxlo"][0]
            got[:, 0] = bed["boxlo"][0] + np.mod(got[:, 0] - bed["boxlo"][0], L)
            refx = a["x"].copy()
            refx[:, 0] = bed["boxlo"][0] + np.mod(refx[:, 0] - bed["boxlo"][0], L)
            assert np.max(np.abs(got - refx)) <= 1e-12
        else:
            assert dc.rel_err(got, a[k]) <= 1e-9, k
    assert all(int(p["rebuilds"]) >= 3 for p in parts)
    hb = {}
    for p in parts:
        for (i, j), s in zip(p["hk"], p["hv"]):
            hb.setdefault((int(i), int(j)), s)     # cross-slab pairs are held by both owners
    assert set(hb) == set(ha)
    sa = np.array([ha[k] for k in sorted(ha)]); sb = np.array([hb[k] for k in sorted(ha)])
    assert dc.rel_err(sb, sa) <= 1e-9


def test_decomposed_cohesive_lubricate_matches_single_domain():
    """Config C5's physics on two slabs: polydisperse grains, fix cohesive, hybrid/overlay lubricate/poly with
    flagVF = flagfld = 1.  The FLD resistances R0 / RT0 need the particle volume of ALL ranks
    (MPI_Allreduce, pair_lubricate_poly.cpp:540-543), the ghost cutoff the largest radius of all ranks and the
    lubrication cutoff; the decomposed run must reproduce the single-domain oracle through rebuilds + migration."""
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from tests import dem_cases as dc
    steps = (40, 40)
    bed, cfg = _case(True, physics="c5")
    ref = dc.make_oracle(bed, cfg)
    ref.setup()
    for n in steps:
        ref.run(n)
    a = ref.get()
    assert ref.nbuilds >= 3
    # the FLD drag really acts (otherwise a wrong volume fraction would go unnoticed): compare with flagVF = 0
    novf = dc.make_oracle(bed, dict(cfg, lub=cfg["lub"][:6] + (0,)))
    withvf = dc.make_oracle(bed, cfg)
    novf.setup(); withvf.setup()
    assert dc.rel_err(novf.get()["f"], withvf.get()["f"]) > 1e-6
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_worker, args=(2, _free_port(), out, True, steps, "fused", "c5"), nprocs=2, join=True)
        parts = [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(2)]
    tag = np.concatenate([p["tag"] for p in parts])
    assert len(tag) == bed["n"] and len(np.unique(tag)) == bed["n"]
    order = np.argsort(tag)
    L = bed["boxhi"][0] - bed["boxlo"][0]
    for k in ("x", "v", "omega", "f", "torque"):
        got = np.concatenate([p[k] for p in parts])[order]
        want = a[k].copy()
        if k == "x":
            got[:, 0] = np.mod(got[:, 0] - bed["boxlo"][0], L); want[:, 0] = np.mod(want[:, 0] - bed["boxlo"][0], L)
            assert np.max(np.abs(got - want)) <= 1e-12
        else:
            assert dc.rel_err(got, want) <= 1e-9, k
    assert all(int(p["rebuilds"]) >= 3 for p in parts)
