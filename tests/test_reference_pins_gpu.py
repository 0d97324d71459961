"""The HIP kernels against numbers computed by the reference's own source lines (tests/golden/reference_pins.json, see
tests/test_reference_pins.py for the CPU half): the stand-alone PairStyle / FixStyle / dragModel entry points (sfk_*) run
on exactly the LAMMPS-shaped inputs the reference's loops were executed on.  Tolerance: 1e-12 of the largest component
(the kernels re-associate a few products and use the hardware's reciprocal / rsqrt seeds)."""
import ctypes as C

import numpy as np
import pytest

from tests import dem_cases as dc
from tests.test_reference_pins import CLOUD_KEY, LUB_KEY, PINS, WALL_KEY, _vunhex, csr, unhex

pytestmark = pytest.mark.gpu


def _t(a, dtype=None):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a, dtype=dtype)).cuda()


@pytest.mark.parametrize("k", range(len(PINS["pair_gran_hertzFix_history.cpp:109-286"])))
def test_hip_hertzfix_history_equals_the_reference_lines(k):
    import torch
    import sedifoam_amd
    from sedifoam_amd._lib import GranParams
    S = sedifoam_amd.lib()
    c = PINS["pair_gran_hertzFix_history.cpp:109-286"][k]
    I, O = c["inp"], c["out"]
    pg = GranParams()
    assert S.sfk_gran_settings(C.byref(pg), I["kn"], 0, I["kt"], I["gamman"], 1, 0.0, I["xmu"], 1, 1.0) == 0
    n, nlocal = I["n"], I["nlocal"]
    first, jl = csr(I["firstneigh"], nlocal)
    touch = np.array([t for i in range(nlocal) for t in I["touch"][i]] or [0], dtype=np.int32)
    shear = np.array([s for i in range(nlocal) for s in I["shear"][i]] or [0.0]).reshape(-1, 3)
    d = dict(x=_t(I["x"], np.float64), v=_t(I["v"], np.float64), w=_t(I["omega"], np.float64),
             r=_t(I["radius"], np.float64), m=_t(I["rmass"], np.float64), mask=_t(I["mask"], np.int32),
             ilist=_t(np.arange(nlocal), np.int32))
    dfirst, djl, dtouch, dshear = _t(first), _t(jl), _t(touch), _t(shear)
    df = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
    dtq = torch.zeros_like(df)
    if "mass_rigid" in I:   # the fix_rigid branch (:72-86, 182-185): the per-atom body masses go in as they are
        dmr = _t(I["mass_rigid"], np.float64)
        assert S.sfk_pair_gran_history_compute_rigid(1, C.byref(pg), I["dt"], I["shearupdate"], nlocal, nlocal,
                                                     d["ilist"].data_ptr(), dfirst.data_ptr(), djl.data_ptr(),
                                                     dtouch.data_ptr(), dshear.data_ptr(), d["x"].data_ptr(),
                                                     d["v"].data_ptr(), d["w"].data_ptr(), d["r"].data_ptr(),
                                                     d["m"].data_ptr(), d["mask"].data_ptr(), I["freeze_group_bit"],
                                                     df.data_ptr(), dtq.data_ptr(), dmr.data_ptr(),
                                                     torch.cuda.current_stream().cuda_stream) == 0
    else:
        assert S.sfk_pair_gran_history_compute(1, C.byref(pg), I["dt"], I["shearupdate"], nlocal, nlocal,
                                               d["ilist"].data_ptr(), dfirst.data_ptr(), djl.data_ptr(), dtouch.data_ptr(),
                                               dshear.data_ptr(), d["x"].data_ptr(), d["v"].data_ptr(), d["w"].data_ptr(),
                                               d["r"].data_ptr(), d["m"].data_ptr(), d["mask"].data_ptr(),
                                               I["freeze_group_bit"], df.data_ptr(), dtq.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    ref_touch = np.array([t for i in range(nlocal) for t in O["touch"][i]], dtype=np.int32)
    ref_shear = np.array([float.fromhex(s) for i in range(nlocal) for s in O["shear"][i]]).reshape(-1, 3)
    assert np.array_equal(dtouch.cpu().numpy()[:ref_touch.size], ref_touch)
    assert dc.rel_err(dshear.cpu().numpy()[:len(ref_shear)], ref_shear) <= 1e-12
    assert dc.rel_err(df.cpu().numpy(), unhex(O["f"])) <= 1e-12
    assert dc.rel_err(dtq.cpu().numpy(), unhex(O["torque"])) <= 1e-12


@pytest.mark.parametrize("k", range(len(PINS["fix_cohesive.cpp:161-262"])))
def test_hip_fix_cohesive_equals_the_reference_lines(k):
    import torch
    import sedifoam_amd
    S = sedifoam_amd.lib()
    c = PINS["fix_cohesive.cpp:161-262"][k]
    I, O = c["inp"], c["out"]
    n, nlocal = I["n"], I["nlocal"]
    first, jl = csr(I["firstneigh"], nlocal)
    df = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
    dx, dr, dmask, dil = _t(I["x"], np.float64), _t(I["radius"], np.float64), _t(I["mask"], np.int32), \
        _t(np.arange(nlocal), np.int32)
    dfirst, djl = _t(first), _t(jl)
    assert S.sfk_fix_cohesive_post_force(I["ah"], I["lam"], I["smin"], I["smax"], I["opt"], nlocal, I["newton_pair"],
                                         dil.data_ptr(), dfirst.data_ptr(), djl.data_ptr(), dx.data_ptr(), dr.data_ptr(),
                                         dmask.data_ptr(), I["groupbit"], df.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert dc.rel_err(df.cpu().numpy(), unhex(O["f"])) <= 1e-12


@pytest.mark.parametrize("k", range(len(PINS[LUB_KEY])))
def test_hip_lubricate_poly_equals_the_reference_lines(k):
    import torch
    import sedifoam_amd
    from sedifoam_amd._lib import LubParams
    S = sedifoam_amd.lib()
    c = PINS[LUB_KEY][k]
    I, O = c["inp"], c["out"]
    n, nlocal = I["n"], I["nlocal"]
    lp = LubParams(I["mu"], I["flaglog"], I["flagfld"], I["flagHI"], I["flagVF"], I["cut_inner"], I["cut_global"],
                   float.fromhex(O["R0"]), float.fromhex(O["RT0"]), float.fromhex(O["RS0"]), 1.0)
    first, jl = csr(I["firstneigh"], nlocal)
    d = dict(x=_t(I["x"], np.float64), v=_t(I["v"], np.float64), w=_t(I["omega"], np.float64),
             r=_t(I["radius"], np.float64), ilist=_t(np.arange(nlocal), np.int32))
    dfirst, djl = _t(first), _t(jl)
    df = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
    dtq = torch.zeros_like(df)
    assert S.sfk_pair_lubricate_poly_compute(C.byref(lp), nlocal, d["ilist"].data_ptr(), dfirst.data_ptr(),
                                             djl.data_ptr(), d["x"].data_ptr(), d["v"].data_ptr(), d["w"].data_ptr(),
                                             d["r"].data_ptr(), df.data_ptr(), dtq.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert dc.rel_err_nan(df.cpu().numpy(), unhex(O["f"])) <= 1e-11
    assert dc.rel_err_nan(dtq.cpu().numpy(), unhex(O["torque"])) <= 1e-11


@pytest.mark.parametrize("k", range(len(PINS["fix_fluid_drag.cpp:143-163"])))
def test_hip_fix_fluid_drag_equals_the_reference_lines(k):
    import torch
    import sedifoam_amd
    S = sedifoam_amd.lib()
    c = PINS["fix_fluid_drag.cpp:143-163"][k]
    I, O = c["inp"], c["out"]
    n = I["n"]
    dv, dm, dr, dmask = _t(I["v"], np.float64), _t(I["rmass"], np.float64), _t(I["radius"], np.float64), \
        _t(I["mask"], np.int32)
    dfd, ddu, dvo, df = _t(I["ffluiddrag"], np.float64), _t(I["DuDt"], np.float64), _t(I["vOld"], np.float64), \
        _t(I["f"], np.float64)
    assert S.sfk_fix_fluid_drag_post_force(n, I["dt"], I["carrier_rho"], dv.data_ptr(), dm.data_ptr(), dr.data_ptr(),
                                           dmask.data_ptr(), I["groupbit"], dfd.data_ptr(), ddu.data_ptr(),
                                           dvo.data_ptr(), df.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert dc.rel_err(df.cpu().numpy(), unhex(O["f"])) <= 1e-13
    assert np.array_equal(dvo.cpu().numpy(), unhex(O["vOld"]))


@pytest.mark.parametrize("k", range(len(PINS[WALL_KEY])))
def test_hip_fix_wall_granfix_equals_the_reference_lines(k):
    """the HIP wall law (the contact-law functions of the sub-step kernel behind sfk_fix_wall_granfix_post_force) on the
    inputs fix_wall_granFix.cpp:286-344 + the three laws of :361-678 were executed on: x / y / z plane pairs, hooke,
    hooke/history, hertz/history, shearupdate 0 and 1, atoms outside the fix's group, history of atoms that left the wall"""
    import torch
    import sedifoam_amd
    S = sedifoam_amd.lib()
    from sedifoam_amd._lib import GranParams
    c = PINS[WALL_KEY][k]
    I, O = c["inp"], c["out"]
    n = I["n"]
    p = GranParams()
    assert S.sfk_gran_settings(C.byref(p), I["kn"], 0, I["kt"], I["gamman"], 0, I["gammat"], I["xmu"], 1, 1.0) == 0
    dx, dv, dw = _t(I["x"], np.float64), _t(I["v"], np.float64), _t(I["omega"], np.float64)
    dr, dm, dmask = _t(I["radius"], np.float64), _t(I["rmass"], np.float64), _t(I["mask"], np.int32)
    dsh = _t(I["shear"], np.float64)
    df, dtq = torch.zeros((n, 3), dtype=torch.float64, device="cuda"), torch.zeros((n, 3), dtype=torch.float64, device="cuda")
    assert S.sfk_fix_wall_granfix_post_force(I["pairstyle"], C.byref(p), I["wallstyle"], I["lo"], I["hi"], 0.0, I["dt"],
                                             I["shearupdate"], n, dx.data_ptr(), dv.data_ptr(), dw.data_ptr(),
                                             dr.data_ptr(), dm.data_ptr(), dmask.data_ptr(), I["groupbit"],
                                             dsh.data_ptr(), df.data_ptr(), dtq.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream) == 0, S.sf_last_error()
    torch.cuda.synchronize()
    ref_f = unhex(O["f"])
    assert np.count_nonzero(ref_f[:, I["wallstyle"]]) >= 20
    assert dc.rel_err(df.cpu().numpy(), ref_f) <= 1e-12
    assert dc.rel_err(dtq.cpu().numpy(), unhex(O["torque"])) <= 1e-12
    if I["pairstyle"] != 0:
        assert dc.rel_err(dsh.cpu().numpy(), unhex(O["shear"])) <= 1e-12


@pytest.mark.parametrize("key,model", [("ErgunWenYu.C:104-132", 0), ("SyamlalOBrien.C:105-143", 1)])
def test_hip_drag_model_jd_equals_the_reference_lines(key, model):
    import torch
    import sedifoam_amd
    S = sedifoam_amd.lib()
    for c in PINS[key]:
        I, O = c["inp"], c["out"]
        n = I["n"]
        dU, da, dp_ = _t(I["Ur"], np.float64), _t(I["alpha"], np.float64), _t(I["pd"], np.float64)
        djd = torch.zeros(n, dtype=torch.float64, device="cuda")
        assert S.sfk_drag_model_jd(model, n, dU.data_ptr(), da.data_ptr(), dp_.data_ptr(), I["nuf"], I["rhof"],
                                   djd.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        ref, got = unhex(O["Jd"]), djd.cpu().numpy()
        # per element: the values span 30 decades (Re clamps to ROOTVSMALL for a particle at rest)
        ok = np.abs(got - ref) <= 1e-9 * np.abs(ref) if model == 1 else np.abs(got - ref) <= 1e-12 * np.abs(ref)
        assert ok.all(), (key, np.max(np.abs(got - ref) / np.abs(ref)))


# the cloud cases the HIP cloud can be walked through without a DEM step in between: one CFD step, no force that needs the
# previous step's particle velocity (added mass, history force)
_CLOUD_HIP = [k for k, c in enumerate(PINS[CLOUD_KEY])
              if c["inp"]["n_steps"] == 1 and not any(c["inp"]["flags"].get(f) for f in
                                                      ("particleAddedMass", "particleHistoryForce", "particleLift",
                                                       "lubricationForce", "particleBuoyancy"))]


@pytest.mark.parametrize("k", _CLOUD_HIP)
def test_hip_cloud_kernels_equal_the_reference_lines(k):
    """the HIP cloud (sf_cloud_*: cell owner, particleToEulerianField, the drag closure + force assembly of
    updateDragOnParticles with the inlet override, calcTcFields) on the inputs enhancedCloud.C:41-108, 129-311, 318-439,
    913-979 were executed on line by line: gamma, Ue, Jd, the force handed to the DEM, Asrc, Omega"""
    from sedifoam_amd import enhancedCloud
    c = PINS[CLOUD_KEY][k]
    I, O = c["inp"], c["out"][0]
    assert len(_CLOUD_HIP) >= 3
    n = I["n"]
    mesh_n = np.array(I["mesh_n"], np.int32)
    dx = np.full(3, I["dx"])
    pos = np.array(I["pos"])
    bed = dict(x=pos, v=np.array(I["U"][1]), diameter=np.array(I["d"]), density=np.full(n, I["rho"]),
               boxlo=np.zeros(3), boxhi=mesh_n * dx, periodic=(0, 0, 0), n=n)
    cfg = dict(pair="hertz", kn=1.0e7, gamman=0.5, xmu=0.4, g=9.81, dt=1.0e-6, skin=0.25e-3, walls=[])
    lmp = dc.make_hip(bed, cfg)
    cloudDict = dict(dragModel={0: "ErgunWenYu", 1: "SyamlalOBrien"}[I["model"]], subCycles=1, g=tuple(I["gravity"]),
                     **{f: bool(v) for f, v in I["flags"].items()})
    if I["inlet"]:
        cloudDict.update(I["inlet"])
    cloud = enhancedCloud(lmp, np.zeros(3), dx, mesh_n, cloudDict, dict(rhob=I["rhob"], nub=I["nub"]), I["deltaT"])
    cloud.setFluid(Uf=np.array(I["Uf"]), DDtUf=np.array(I["DDtUf"]), gradp=np.array(I["gradp"]), curlU=np.array(I["curlU"]))
    cloud._phase(0)        # the next time step
    cloud._phase(1)        # updateParticleAlpha / Ur, Jd, updateDragOnParticles
    cloud.calcTcFields()
    P = cloud.particles()
    o = np.argsort(P["tag"])
    assert np.array_equal(P["tag"][o], np.arange(1, n + 1))
    assert np.array_equal(P["cell"][o], np.array(I["cell"]))       # (the owner the reference's tracking would give)
    g, ue, asrc, om = cloud._fields()
    assert dc.rel_err(g, unhex(O["gamma"])) <= 1e-12
    assert dc.rel_err(ue, _vunhex(O["Ue"])) <= 1e-12
    assert dc.rel_err(P["Jd"][o], unhex(O["Jd"])) <= 1e-12
    assert dc.rel_err(P["pDrag"][o], _vunhex(O["pDrag"])) <= 1e-12
    assert dc.rel_err(asrc, _vunhex(O["Asrc"])) <= 1e-12
    assert dc.rel_err(om, unhex(O["Omega"])) <= 1e-12
    if I["inlet"]:
        assert np.count_nonzero(np.any(_vunhex(O["pDrag"]) != 0.0, axis=1)) > 10
