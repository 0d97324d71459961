#!/usr/bin/env python3
"""bench.py -- particle-DEM-substeps/s of the sediFoam hot path on MI355X.

One "step" = one `lammps_step(S)` call (S = 50 DEM sub-steps, BASELINE.json configs[2..3]) of the fused
Hertz-history contact / fix fdrag / wall / gravity / nve-sphere kernel over a synthetic 1 M-particle
monodisperse Hertz packing, particle state resident in HBM before timing starts.
With --gpus N (launched through torch.distributed.run, one rank per GPU) the SAME 1 M-particle bed is split into N
spatial domains with a ghost-particle halo over RCCL (BASELINE config C4: strong scaling, the `value` of the line);
the weak-scaling run (every rank one 1 M slab of a channel N times as long) goes into the side object `weak_scaling`.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (k_substep): algorithmic bytes
per launch (284 + 52*K_half bytes per particle-substep) / its mean duration from HIP
events on the engine's own stream.  `cpu_baseline` times the CPU oracle (a port of the reference's
algorithm; the reference itself needs LAMMPS + OpenFOAM, absent here) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
PMC_SUMMARIES = ("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json")   # committed PMC passes, newest first (tests/profile_round.sh)


def build_engine(bed, script):
    from sedifoam_amd import Lammps
    lmp = Lammps()
    lmp.set_box(bed["boxlo"], bed["boxhi"])
    lmp.create_atoms(bed["x"], bed["diameter"], bed["density"], v=bed["v"])
    for line in script:
        lmp.command(line)
    return lmp


def _synthetic():
    """sedifoam_amd/synthetic.py without importing the package (the CPU workers need neither torch nor the HIP library)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sf_synthetic", os.path.join(ROOT, "sedifoam_amd", "synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def cpu_baseline(ncells, script_kw, substeps, seed=12345 + 3, keep_state=False):
    """Oracle (CPU port of the reference algorithm) on the same kind of bed: particle-substeps/s.  keep_state: also the
    oracle's x, v, omega (sorted by tag) after those sub-steps -- the parity leg compares them with the GPU's -- and the
    oracle itself with its bed (the coupled CPU step continues from there)."""
    from oracle import binding as ob
    synthetic = _synthetic()
    bed = synthetic.fcc_bed(ncells, seed=seed)
    r = 0.5 * bed["diameter"]
    m = 4.0 * np.pi / 3.0 * r ** 3 * bed["density"]
    dem = ob.OracleDem(bed["x"], r, m, bed["boxlo"], bed["boxhi"], periodic=bed["periodic"], v=bed["v"])
    dem.pair_gran("hertz", script_kw["kn"], None, script_kw["gamman"], None, script_kw["xmu"], 1)
    dem.fix_gravity(script_kw["g"], 0.0, -1.0, 0.0)
    dem.fix_fdrag(0.0)
    dem.fix_wall(1, float(bed["boxlo"][1]), float(bed["boxhi"][1]), script_kw["kn"], None, script_kw["gamman"],
                 None, script_kw["xmu"], 1)
    dem.neighbor(script_kw["skin_d"] * 1.0e-3)
    dem.timestep(script_kw["dt"])
    dem.setup()
    t0 = time.perf_counter()
    dem.run(substeps)
    dt = time.perf_counter() - t0
    if keep_state:
        return bed["n"] * substeps / dt, bed["n"], dt, dem.get(), dem, bed
    return bed["n"] * substeps / dt, bed["n"], dt


def cpu_coupled_step(dem, bed, mesh_n, substeps, band, steps):
    """ONE coupled CFD-DEM step of the reference's algorithm on the CPU (oracle/orc_cloud.c + the oracle's DEM loop), the
    same step the GPU's `coupled_steps_per_s` times -- enhancedCloud::evolve() with subCycles 1 (UfSmoothed, drag closure
    + assembly, lammps_put_local_info, `substeps` DEM sub-steps, cell owner, particle -> Eulerian scatter with diffusion
    smoothing) and calcTcFields() -- with the wall clock split into the reference's buckets (writeCPUTime.H:1-19)."""
    import ctypes as C
    from oracle import binding as ob
    L = ob.lib()
    mesh_n = np.array(mesh_n, np.int32)
    origin = np.array(bed["boxlo"], np.float64)
    dxm = (np.array(bed["boxhi"], np.float64) - origin) / mesh_n
    ncells = int(np.prod(mesh_n))
    sm = ob.Smooth()
    sm.n = (C.c_int * 3)(*[int(k) for k in mesh_n]); sm.dx = (C.c_double * 3)(*dxm); sm.D = (C.c_double * 3)(1.0, 1.0, 1.0)
    sm.band = band; sm.steps = steps; sm.UfSmooth = sm.UpSmooth = sm.dragSmooth = sm.alphaSmooth = 1
    smp = C.byref(sm)
    fl = ob.CloudFlags()
    fl.particleDrag = 1; fl.particlePressureGrad = 1
    fl.gravity = (C.c_double * 3)(0.0, -9.81, 0.0); fl.rhob = 1000.0; fl.nub = 1.0e-6; fl.deltaT = substeps * KW["dt"]
    st = dem.get()
    n = st["x"].shape[0]
    d = np.ascontiguousarray(bed["diameter"], np.float64)
    V = np.full(ncells, float(np.prod(dxm)))
    Uf = np.tile([0.0, 0.05, 0.0], (ncells, 1)); gradp = np.tile([0.0, -9810.0, 0.0], (ncells, 1))
    zc = np.zeros((ncells, 3))
    gamma = np.zeros(ncells); Ue = np.zeros((ncells, 3)); cell = np.zeros(n, np.int32); UfS = np.zeros((ncells, 3))
    L.orc_cell_owner(n, ob.P(st["x"]), ob.P(origin), ob.P(dxm), ob.P(mesh_n), ob.P(cell))
    L.orc_particle_to_eulerian_smooth(n, ob.P(cell), ob.P(d), ob.P(st["v"]), ncells, ob.P(V), smp, ob.P(gamma), ob.P(Ue))
    tb = {}

    def lap(name, t):
        tb[name] = tb.get(name, 0.0) + time.perf_counter() - t
    t_ev = time.perf_counter()
    t = time.perf_counter()
    L.orc_uf_smoothed(ncells, ob.P(Uf), ob.P(gamma), smp, ob.P(UfS))                  # enhancedCloud.C:675-690
    Uri = np.zeros((n, 3)); mag = np.zeros(n); Jd = np.zeros(n); pDrag = np.zeros((n, 3)); pDuDt = np.zeros((n, 3))
    sumFb = np.zeros((n, 3)); n0 = np.zeros(n)
    L.orc_drag_on_particles_hist(C.byref(fl), 0, n, ob.P(cell), ob.P(st["x"]), ob.P(d), ob.P(st["v"]), ob.P(st["v"]),
                                 ob.P(gamma), ob.P(UfS), ob.P(gradp), ob.P(zc), ob.P(zc), -1, ob.P(UfS), ob.P(sumFb),
                                 ob.P(n0), ob.P(Uri), ob.P(mag), ob.P(Jd), ob.P(pDrag), ob.P(pDuDt))
    dem.put_fdrag(pDrag, st["tag"])
    lap("foam->lammps (drag closure + assembly)", t)
    t = time.perf_counter()
    dem.run(substeps)
    lap("lammps (%d sub-steps)" % substeps, t)
    t = time.perf_counter()
    st = dem.get()
    L.orc_cell_owner(n, ob.P(st["x"]), ob.P(origin), ob.P(dxm), ob.P(mesh_n), ob.P(cell))
    L.orc_particle_to_eulerian_smooth(n, ob.P(cell), ob.P(d), ob.P(st["v"]), ncells, ob.P(V), smp, ob.P(gamma), ob.P(Ue))
    lap("particle move (cell owner + scatter + smoothing)", t)
    tb["evolve"] = time.perf_counter() - t_ev
    t = time.perf_counter()
    gcap = np.minimum(gamma, 0.65)                                                     # liftDragCoeffs.H:6-14
    Ur = np.linalg.norm(UfS[cell] - st["v"], axis=1)
    L.orc_ergun_wenyu_jd(n, ob.P(Ur), ob.P(np.ascontiguousarray(gcap[cell])), ob.P(d), 1.0e-6, 1000.0, ob.P(Jd))
    Asrc = np.zeros((ncells, 3)); Omega = np.ones(ncells)
    L.orc_calc_tc_fields_smooth(n, ob.P(cell), ob.P(d), ob.P(st["v"]), ob.P(Jd), ncells, ob.P(V), ob.P(gcap), ob.P(UfS),
                                smp, ob.P(Asrc), ob.P(Omega))
    lap("calcTcField", t)
    total = tb["evolve"] + tb["calcTcField"]
    return total, {k: 1e3 * v for k, v in tb.items()}


KW = dict(kn=1.0e7, gamman=0.5, xmu=0.4, dt=1.0e-6, skin_d=0.25, g=9.81)
STAGE = ["start"]   # what the run was doing, for the `error` line of a run that does not come up (N > 1 above all)


def error_line(args_gpus, steps, warmup, msg):
    """the ONE JSON line of a run that failed before it had a number: same keys, value null, what failed and where"""
    return json.dumps({"metric": "particle-DEM-substeps/sec", "value": None, "unit": "particle-substeps/s", "n_gpus": args_gpus,
                       "steps": steps, "warmup": warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
                       "vs_baseline": None, "dtype": "f64", "data": "synthetic", "error": str(msg)[:600], "stage": STAGE[0],
                       "rank": int(os.environ.get("RANK", "0")), "world": int(os.environ.get("WORLD_SIZE", "1"))})
FLUIDISED = dict(jitter=0.3, spacing=1.1)   # --bed fluidised


def self_launch(n, steps, warmup):
    """`python3 bench.py --gpus N` without a launcher: start the N ranks here -- the same command line, RANK / LOCAL_RANK /
    WORLD_SIZE in the environment, rendezvous through a FileStore in a directory of this run (no port to collide on) -- pass
    rank 0's standard output through, and return the job's exit code: 0 only when every rank returned 0.  A rank that dies takes
    the others with it after a short grace (they would wait in a collective for ever); if rank 0 has not printed its JSON
    line by then, the `error` line is printed here, so the caller always gets exactly one parseable line."""
    import shutil
    import subprocess
    import tempfile
    import threading
    d = tempfile.mkdtemp(prefix="sf_bench_rdzv_")
    env = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), SF_BENCH_RDZV="file://" + os.path.join(d, "store"),
               MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    got_line = [False]

    def forward():
        for line in procs[0].stdout:
            if line.lstrip().startswith("{") and '"metric"' in line:
                got_line[0] = True
            sys.stdout.write(line)
            sys.stdout.flush()
    th = threading.Thread(target=forward, daemon=True)
    th.start()
    rc, first_bad, t_bad = 0, None, None
    try:
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad and first_bad is None:
                first_bad, t_bad = bad[0], time.time()
            if first_bad is not None and time.time() - t_bad > 20.0:   # (rank 0's own handler gets its chance to print first)
                for p in procs:
                    if p.poll() is None:
                        p.kill()      # (exactly the processes started above)
            time.sleep(0.2)
        th.join(10)
        codes = [p.returncode for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c != 0]
        if first_bad is None and bad:
            first_bad = bad[0]
        rc = 0 if not bad else (first_bad[1] if first_bad[1] and first_bad[1] > 0 else 1)
        if not got_line[0]:
            STAGE[0] = "self-launch of %d ranks" % n
            print(error_line(n, steps, warmup, "no result line from rank 0; exit codes by rank: %s" % codes))
            sys.stdout.flush()
            rc = rc or 1
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(d, ignore_errors=True)
    return rc


def cpu_worker(argv):
    """`bench.py --cpu-worker NPART SUBSTEPS SEED`: one reference-style rank (its own slab of the bed) on one core."""
    npart, sub, seed = int(argv[0]), int(argv[1]), int(argv[2])
    v, n_s, secs = cpu_baseline(_synthetic().fcc_cells_for(npart), KW, sub, seed=seed)
    print(json.dumps({"value": v, "n": int(n_s), "secs": secs}))


def _usable_cores():
    """cores this process may really use: the affinity mask, capped by the cgroup CPU quota of the container"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]           # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())           # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline_all_cores(npart, sub):
    """What `mpirun -np <cores>` of the reference does on this host, without its halo traffic: one oracle process per
    core, each with its own slab of `npart` particles (weak, like the GPU ranks); throughput = sum over processes of
    particles x sub-steps / the slowest process's run time (setup and list build are not timed, as on the GPU)."""
    import subprocess
    cores = _usable_cores()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(npart), str(sub),
                               str(777 + c)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for c in range(cores)]
    res = []
    for pr in procs:
        out_s, _ = pr.communicate(timeout=600)
        if pr.returncode == 0:
            res.append(json.loads(out_s.strip().splitlines()[-1]))
    if not res:
        return None
    slowest = max(r["secs"] for r in res)
    return {"value": sum(r["n"] for r in res) * sub / slowest, "unit": "particle-substeps/s", "cores": len(res),
            "kind": "port", "sample": "%d independent oracle processes (one per core), %d particles x %d sub-steps each, "
                                      "slowest %.1f s; no halo exchange between them" % (len(res), res[0]["n"], sub, slowest)}


# ---- BASELINE.json configs[1], [2], [4] (C2, C3, C5) next to the headline (configs[3] = C4): the `configs` object ----
C5_LUB = (1.0e-3, 1, 1, 1.001e-3, 1.1e-3, 1, 1)        # pair lubricate/poly mu flaglog flagfld cut_inner cut_global flagHI flagVF
C5_COHESIVE = (1.0e-13, 1.0e-7, 1.0e-7, 1.0e-4, 1)     # fix cohesive ah lam smin smax opt
C5W_LUB = (1.0e-3, 1, 0, 1.001 * 1.5e-3, 1.1 * 1.5e-3, 1, 1)   # flagfld 0, cutoffs in units of the largest pair
C5W_SUBSTEPS = 10   # sub-steps per step of configs.C5_wide (1 warm-up + K timed steps: 40 sub-steps in all by default)
C5W_COHESIVE = (1.0e-20, 1.0e-7, 1.0e-9, 1.0e-4, 1)           # ah 1e-20, smin 1e-9, smax 0.1 d


def config_cases(synthetic):
    """name -> (bed, cfg, coupled mesh or None, label).  cfg is the dictionary tests/dem_cases.py turns into the oracle's
    set-up; config_script() below writes the same thing as in.lammps lines for the GPU engine."""
    out = {}
    # C2: 10 k monodisperse spheres standing free on the floor of a closed (32 x 3.5 d)^3 box, 32^3 mesh
    nc = synthetic.fcc_cells_for(10000)
    bed = synthetic.fcc_bed(nc, seed=12345 + 1, vmax=0.01)
    box = 32 * 3.5e-3
    bed["x"][:, 0] += 0.5 * (box - nc[0] * bed["edge"])
    bed["x"][:, 2] += 0.5 * (box - nc[2] * bed["edge"])
    bed["periodic"] = (0, 0, 0)
    bed["boxhi"] = np.array([box, box, box])
    cfg = dict(kn=KW["kn"], gamman=KW["gamman"], xmu=KW["xmu"], g=KW["g"], dt=KW["dt"], skin=KW["skin_d"] * 1.0e-3,
               walls=[(1, 0.0, box), (0, 0.0, box), (2, 0.0, box)])
    out["C2"] = (bed, cfg, (32, 32, 32), "%d monodisperse spheres, Hertz history + ErgunWenYu drag, 32^3 mesh (closed box, "
                                          "three wall pairs), 50 DEM sub-steps per CFD step" % bed["n"])
    # C3: 100 k-grain fluidised (loose, disordered) bed, 50 sub-steps per CFD step
    bed = synthetic.fcc_bed(synthetic.fcc_cells_for(100000), seed=12345 + 2, **FLUIDISED)
    cfg = dict(kn=KW["kn"], gamman=KW["gamman"], xmu=KW["xmu"], g=KW["g"], dt=KW["dt"], skin=KW["skin_d"] * 1.0e-3,
               walls=[(1, float(bed["boxlo"][1]), float(bed["boxhi"][1]))])
    mesh = tuple(int(k) for k in np.clip(((bed["boxhi"] - bed["boxlo"]) / 3.3e-3).astype(int), 1, 32))
    out["C3"] = (bed, cfg, mesh, "%d-grain fluidised bed (FCC sites at spacing 1.1 d, jitter 0.3 d), Hertz history + ErgunWenYu "
                                 "drag, %dx%dx%d mesh, 50 DEM sub-steps per CFD step" % ((bed["n"],) + mesh))
    # C5: 500 k polydisperse grains, hybrid/overlay gran/hertzFix/history + lubricate/poly, fix cohesive (periodic box)
    nc = synthetic.fcc_cells_for(500000)
    bed = synthetic.fcc_bed(nc, seed=15, vmax=0.05, poly=(0.85e-3, 1.0e-3), spacing=0.95)
    bed["boxhi"][1] = nc[1] * bed["edge"]
    bed["x"][:, 1] %= bed["boxhi"][1]
    bed["periodic"] = (1, 1, 1)
    cfg = dict(kn=KW["kn"], gamman=KW["gamman"], xmu=KW["xmu"], g=0.0, dt=KW["dt"], skin=0.06e-3, walls=[],
               cohesive=C5_COHESIVE, lub=C5_LUB)
    out["C5"] = (bed, cfg, None, "%d polydisperse grains (d = 0.85-1.0 mm), pair hybrid/overlay gran/hertzFix/history + "
                                 "lubricate/poly, fix cohesive, periodic box, 50 DEM sub-steps per step" % bed["n"])
    # C5 on the size distribution SURVEY.md 8(d) fixes: d ~ U(0.5, 1.5) mm (ratio 3), fix cohesive 1e-20 1e-7 1e-9 0.1 d 1,
    # lubricate/poly 1e-3 1 0 with the inner / outer cutoff 1.001 / 1.1 of the largest pair; a dense disordered periodic bed
    # grown by the engine itself (synthetic.grown_poly_bed: deterministic)
    bed = synthetic.grown_poly_bed(500000, seed=15, vmax=0.05)
    cfg = dict(kn=KW["kn"], gamman=KW["gamman"], xmu=KW["xmu"], g=0.0, dt=KW["dt"], skin=0.06e-3, walls=[],
               cohesive=C5W_COHESIVE, lub=C5W_LUB)
    out["C5_wide"] = (bed, cfg, None, "%d polydisperse grains d ~ U(0.5, 1.5) mm (SURVEY.md 8d), dense disordered periodic bed "
                                      "(solid fraction 0.58, grown), pair hybrid/overlay gran/hertzFix/history + lubricate/poly "
                                      "1e-3 1 0 1.5015e-3 1.65e-3, fix cohesive 1e-20 1e-7 1e-9 1e-4 1, %d DEM sub-steps per "
                                      "step (the reference's log series with its h_sep = 100 (ri + rj) edit is anti-damped at "
                                      "this size ratio: oracle and HIP alike multiply the velocities by ~100 per 50 sub-steps, "
                                      "profiles/r06_README.md section 4 -- the timed region ends before the bed flies apart)"
                                      % (bed["n"], C5W_SUBSTEPS))
    return out


def config_script(bed, cfg):
    gran = "gran/hertzFix/history %.17g NULL %.17g NULL %.17g 1" % (cfg["kn"], cfg["gamman"], cfg["xmu"])
    pair = gran if not cfg.get("lub") else ("hybrid/overlay %s lubricate/poly %.17g %d %d %.17g %.17g %d %d"
                                            % ((gran,) + tuple(cfg["lub"])))
    lines = ["atom_style sphere", "boundary %s %s %s" % tuple("p" if q else "f" for q in bed["periodic"]), "newton off",
             "communicate single vel yes", "neighbor %.17g bin" % cfg["skin"], "neigh_modify delay 0",
             "pair_style " + pair, "pair_coeff * *", "timestep %.17g" % cfg["dt"], "fix 1 all nve/sphere",
             "fix 2 all gravity %.17g vector 0 -1 0" % cfg["g"], "fix 3 all fdrag"]
    for k, (dim, lo, hi) in enumerate(cfg["walls"]):
        lines.append("fix w%d all wall/granFix %.17g NULL %.17g NULL %.17g 1 %splane %.17g %.17g"
                     % (k, cfg["kn"], cfg["gamman"], cfg["xmu"], "xyz"[dim], lo, hi))
    if cfg.get("cohesive"):
        lines.append("fix coh all cohesive %.17g %.17g %.17g %.17g %d" % tuple(cfg["cohesive"]))
    return lines


def config_cpu_leg(bed, cfg, substeps):
    """the oracle on the same bed and script (tests/dem_cases.py builds it from the same dictionary), `substeps` DEM sub-steps
    after setup, single thread: the bounded CPU baseline of one named configuration"""
    from tests import dem_cases as dc
    orc = dc.make_oracle(bed, dict(cfg, pair="hertz"))
    orc.setup()
    t0 = time.perf_counter()
    orc.run(substeps)
    secs = time.perf_counter() - t0
    return {"value": bed["n"] * substeps / secs, "unit": "particle-substeps/s", "cores": 1, "kind": "port",
            "sample": "the same %d-particle bed and script, %d sub-steps after setup, %.2f s, oracle/ (C, gcc -O2) single "
                      "thread" % (bed["n"], substeps, secs)}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--particles", type=int, default=1000000, help="particles per GPU")
    ap.add_argument("--substeps", type=int, default=50, help="DEM sub-steps per step (per CFD step)")
    ap.add_argument("--jitter", type=float, default=None, help="sensitivity runs: uniform position jitter / d (default 0.005)")
    ap.add_argument("--spacing", type=float, default=None, help="sensitivity runs: lattice spacing / d (default 0.98)")
    ap.add_argument("--skin", type=float, default=None, help="sensitivity runs: neighbour skin / d (default 0.25)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-coupled", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true",
                    help="do not sample per-launch HIP events in the timed region (the roofline leg is then empty)")
    ap.add_argument("--slab-driver", action="store_true",
                    help="drive the sub-steps through the multi-rank SlabDriver even at N=1 (self halo)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="oracle sample size (particles), 0 = auto")
    ap.add_argument("--cpu-all-particles", type=int, default=100000,
                    help="particles per process of the all-cores CPU leg (one oracle process per host core), 0 = skip")
    ap.add_argument("--scaling", choices=["both", "weak", "strong"], default="both",
                    help="N > 1: strong = --particles in total, split into N domains (BASELINE config C4: the `value` of "
                         "the JSON line); weak = every rank owns one --particles slab (`value` only with --scaling weak); "
                         "both (default) = strong as `value`, weak next to it as `weak_scaling`.  N = 1: identical")
    ap.add_argument("--decomposition", choices=["auto", "slabs", "bricks"], default="auto",
                    help="N > 1, strong scaling: x-slabs, or bricks of a 3-D processor grid (sedifoam_amd.halo.brick_grid: "
                         "periodic dimensions first, as cubic as N allows: 8 -> 4x1x2, 4 -> 2x1x2 on the channel bed); "
                         "auto = bricks for every N (2 -> 2x1x1: the driver with the direct ghost writes; from 4 ranks on half the bytes per face, two to three links busy at once)")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="N > 1: if the C++ RCCL driver cannot come up, measure the Python loop over torch.distributed "
                         "instead of exiting non-zero (config.decomposition says so)")
    ap.add_argument("--bed", choices=["packed", "fluidised"], default="packed",
                    help="packed = the lattice bed of BASELINE's headline; fluidised = a loose disordered bed (jitter 0.3 d, "
                         "spacing 1.1 d: K_half ~4, a third of the listed neighbours touch, a rebuild every ~11 sub-steps)")
    ap.add_argument("--no-fluidised", action="store_true", help="N = 1: skip the `fluidised_bed` side measurement")
    ap.add_argument("--no-configs", action="store_true",
                    help="N = 1: skip the `configs` object (BASELINE configs C2, C3, C5 next to the headline) and the 2 M bed "
                         "of `roofline.frac_2m`")
    ap.add_argument("--no-parity", action="store_true", help="N = 1: skip the GPU-vs-oracle comparison of the final state")
    ap.add_argument("--one-gpu", action="store_true",
                    help="development: all ranks share GPU 0, halo over gloo through host memory (RCCL refuses two "
                         "ranks on one device); exercises the N > 1 code on a 1-GPU box")
    ap.add_argument("--coupled-replicated", action="store_true",
                    help="--coupled-multi: keep the whole mesh on every rank (three mesh-sized all-reduces per CFD step) "
                         "instead of cutting it by the slab planes")
    ap.add_argument("--coupled-multi", action="store_true",
                    help="with --gpus N > 1 also time coupled steps: enhancedCloud over the decomposed particles, "
                         "whole mesh on every rank, per-cell sums all-reduced")
    ap.add_argument("--watchdog", type=float, default=None,
                    help="N > 1: seconds after which a rank that is still running ends the job with rc 124 (a collective "
                         "that never completes must not hold the node forever; default 1500, 0 = off)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python3 bench.py --gpus N`: launch the N ranks from here (one process per GPU, FileStore rendezvous)
        raise SystemExit(self_launch(args.gpus, args.steps, args.warmup))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d under a launcher with WORLD_SIZE=%d: one rank per GPU, the two must agree"
                         % (args.gpus, world))

    wd = args.watchdog if args.watchdog is not None else (1500.0 if world > 1 else 0.0)
    if wd > 0:
        import threading

        def _expired():
            sys.stderr.write("bench.py: rank %d still running after %.0f s (--watchdog): a collective or a halo exchange "
                             "never completed -- ending the job, rc 124\n" % (rank, wd))
            sys.stderr.flush()
            if rank == 0:   # (the line the driver parses says what hung and where)
                sys.stdout.write(error_line(args.gpus, args.steps, args.warmup,
                                            "watchdog: still running after %.0f s in stage '%s'" % (wd, STAGE[0])) + "\n")
                sys.stdout.flush()
            os._exit(124)
        t = threading.Timer(wd, _expired)
        t.daemon = True
        t.start()

    # every 16th sub-step kernel of the timed region sits between a HIP event pair (`roofline.achieved`): a pair costs the run
    # ~12 us of launch gaps (kernel trace), i.e. 0.4 % of the whole-run value at this stride
    os.environ.setdefault("SF_PROF_STRIDE", "16")

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    if args.one_gpu:
        local_rank = 0
        os.environ.setdefault("SF_HALO_DIRECT_TIMEOUT", "120")   # (ranks sharing ONE GPU wait for each other's time slices)
    torch.cuda.set_device(local_rank)
    dist = None
    # --one-gpu: gloo through host memory from Python, or -- with SF_RCCL_LIB pointing at tests/c_abi/standin_rccl.cpp
    # built as a library -- the C++ driver of the real N > 1 run over that stand-in for librccl
    transport = ("rccl" if os.environ.get("SF_RCCL_LIB") else "host") if args.one_gpu else None
    if world > 1 or os.environ.get("SF_HALO_SELF_COMM", "0") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist_mod
        dist = dist_mod
        STAGE[0] = "torch.distributed.init_process_group (%s, %s:%s)" % ("gloo" if args.one_gpu else "nccl = RCCL",
                                                                          os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"))
        rdzv = os.environ.get("SF_BENCH_RDZV")      # (set by self_launch: a FileStore, no port; else the launcher's env://)
        rdzv_kw = dict(init_method=rdzv, rank=rank, world_size=world) if rdzv else {}
        if args.one_gpu:
            dist.init_process_group("gloo", **rdzv_kw)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), **rdzv_kw)
            # (the first collective brings the RCCL communicator up: fail here, with this stage name, rather than inside a driver)
            STAGE[0] = "first RCCL all-reduce over %d ranks" % world
            t_ = torch.ones(1, device="cuda")
            dist.all_reduce(t_)
            torch.cuda.synchronize()

    from sedifoam_amd import synthetic
    kw = dict(KW, skin_d=args.skin) if args.skin is not None else KW
    ncells = synthetic.fcc_cells_for(args.particles)
    bed_kw = {}
    if args.bed == "fluidised":
        bed_kw = dict(FLUIDISED)
    if args.jitter is not None:
        bed_kw["jitter"] = args.jitter
    if args.spacing is not None:
        bed_kw["spacing"] = args.spacing
    bed = synthetic.fcc_bed(ncells, seed=12345 + 3 + rank, **bed_kw)
    script = synthetic.hertz_script(bed, **kw)
    N = bed["n"]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    exchange_us = [None]
    rebuild_ms = [None]

    def timed_run(lmp):
        """W warm-up + K timed steps of `lammps_step(S)`; returns (elapsed max over ranks, total particles, launches,
        kernel ms, info before, info after)"""
        STAGE[0] = "setup (first list build, halo bring-up) of %s" % type(lmp).__name__
        lmp.setup()
        STAGE[0] = "stepping %s" % type(lmp).__name__
        info0 = lmp.info()
        for _ in range(args.warmup):
            lmp.step(args.substeps)
        lmp.set_profiling(not args.no_kernel_profile)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            lmp.step(args.substeps)
        barrier()
        el = time.perf_counter() - t0
        launches, kernel_ms = lmp.get_profile()
        # decomposed run through the C++ driver: what a forward exchange (vote + ghosts over RCCL, unpack) costs on this
        # rank's stream, HIP events around every 8th one; the slowest rank's mean goes into the line
        xn, xms = lmp.get_exchange_profile() if hasattr(lmp, "get_exchange_profile") else (0, 0.0)
        exchange_us[0] = 1e3 * xms / xn if xn else None
        rn, rms = lmp.get_rebuild_profile() if hasattr(lmp, "get_rebuild_profile") else (0, 0.0)
        rebuild_ms[0] = rms / rn if rn else None
        lmp.set_profiling(False)
        n_own = float(lmp.info().nlocal)
        if dist is not None:
            rdev = "cpu" if args.one_gpu else "cuda"
            t = torch.tensor([el], dtype=torch.float64, device=rdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
            nt = torch.tensor([n_own], dtype=torch.float64, device=rdev)
            dist.all_reduce(nt)
            n_own = float(nt.item())
            xt = torch.tensor([exchange_us[0] if exchange_us[0] is not None else -1.0], dtype=torch.float64, device=rdev)
            dist.all_reduce(xt, op=dist.ReduceOp.MAX)
            exchange_us[0] = float(xt.item()) if xt.item() >= 0 else None
        return el, n_own, launches, kernel_ms, info0, lmp.info()

    fallback_note = [None]
    comm_info = [None]

    grid_used = [None]

    def make_driver(factory, the_bed):
        """the C++ driver over RCCL.  If it cannot come up the run FAILS (rc != 0): a slow number from another code path
        is worse than none.  --allow-fallback: say so loudly and measure the same protocol driven from Python over
        torch.distributed instead."""
        from sedifoam_amd.halo import SlabDriver
        from sedifoam_amd.halo import brick_grid
        err = None
        kw_grid = {}
        STAGE[0] = "creating the C++ halo driver (%s)" % factory
        # (auto: bricks for every N -- 2 x 1 x 1 are the two slabs, cut by the driver that also has the direct ghost writes;
        # the slab driver stays behind --decomposition slabs, and under --one-gpu without the stand-in wire)
        if factory == "from_global_bed" and (args.decomposition == "bricks" or (args.decomposition == "auto" and world >= 2
                                                                                   and not (args.one_gpu and transport != "rccl"))):
            if args.one_gpu and transport != "rccl":
                raise SystemExit("bench.py --one-gpu with bricks needs SF_RCCL_LIB (the brick driver is C++ only)")
            kw_grid = {"grid": brick_grid(world, the_bed)}
            grid_used[0] = kw_grid["grid"]
        try:
            if kw_grid:
                from sedifoam_amd.halo import BrickDriver
                drv = BrickDriver.from_global_bed(the_bed, script, dist, rank, world, kw_grid["grid"])
            else:
                drv = getattr(SlabDriver, factory)(the_bed, script, dist, rank, world, transport=transport)
        except Exception as ex:   # noqa: BLE001
            drv, err = None, ex
        if dist is not None and world > 1:
            rdev = "cpu" if args.one_gpu else "cuda"
            flag = torch.tensor([1.0 if err is not None else 0.0], dtype=torch.float64, device=rdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            failed = flag.item() > 0
        else:
            failed = err is not None
        if not failed:
            if hasattr(drv, "comm_info"):
                comm_info[0] = drv.comm_info()
            return drv
        if not args.allow_fallback or args.one_gpu or transport not in (None, "rccl"):
            raise err if err is not None else RuntimeError("another rank could not create its halo driver")
        sys.stderr.write("bench.py: the C++ RCCL halo driver did not come up (%s) -- --allow-fallback: measuring the Python "
                         "loop over torch.distributed (slower; config.decomposition says so)\n" % (err,))
        fallback_note[0] = "x-slabs, ghost halo driven from Python over torch.distributed (the C++ RCCL driver failed: %s)" % (err,)
        return getattr(SlabDriver, factory)(the_bed, script, dist, rank, world, transport="direct")

    def side_line(el, n_tot, launches_, kernel_ms_, info_, label, scaling):
        o = {"value": n_tot * args.substeps * args.steps / el, "unit": "particle-substeps/s",
             "halo_exchange_us_per_substep": exchange_us[0], "ms_per_step": 1e3 * el / args.steps,
             "particles_total": int(n_tot), "scaling": scaling, "workload": label}
        if launches_:
            kh = info_.npairs_full / 2.0 / max(info_.nlocal, 1)
            o["rank0_kernel_us"] = 1e3 * kernel_ms_ / launches_
            o["rank0_roofline_frac"] = (284.0 + 52.0 * kh) * info_.nlocal / (1e-3 * kernel_ms_ / launches_) / 1e9 / HBM_PEAK_GBS
        return o

    def all_ok(flag):
        """True only if `flag` is true on every rank"""
        if dist is None or world == 1:
            return bool(flag)
        t = torch.tensor([0.0 if flag else 1.0], dtype=torch.float64, device="cpu" if args.one_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item() == 0.0

    trial_ms = [None]   # per 50 sub-steps, of the transport the last parity leg ran on (rank 0's clock between barriers)

    def decomposed_parity():
        """N > 1: the decomposed engine proves itself -- the SAME global bed through setup + 50 sub-steps on the N domains
        (the headline's decomposition and halo transport) and on ONE domain (rank 0's GPU).  Returns (parity dict on rank
        0 / None elsewhere, True if the leg ran through on every rank and agreed)."""
        sub = 50
        STAGE[0] = "parity leg (decomposed vs one domain), SF_HALO_DIRECT=%s" % os.environ.get("SF_HALO_DIRECT")
        gbed = synthetic.fcc_bed(ncells, seed=12345 + 3, **bed_kw)
        mine, mine_builds, direct_on, err = None, 0, 0, None
        trial_ms[0] = None
        try:
            pdrv = make_driver("from_global_bed", gbed)
            pdrv.setup()
            pdrv.step(sub)
            mine = pdrv.e.lmp.get_state()
            mine_builds = int(pdrv.n_rebuilds)
            direct_on = int(pdrv.e.lmp.L.sf_slab_direct_halo(pdrv.e.lmp.ptr)) if grid_used[0] else 0
            # (what this transport costs on THIS node: two more steps of 50 sub-steps between barriers -- the ladder below
            # keeps the faster of two transports that both passed)
            barrier()
            t0 = time.perf_counter()
            pdrv.step(sub)
            pdrv.step(sub)
            barrier()
            trial_ms[0] = 0.5e3 * (time.perf_counter() - t0)
            del pdrv
        except Exception as ex:   # noqa: BLE001  (e.g. a bounded flag wait of the direct ghost writes ran out)
            err = ex
            sys.stderr.write("bench.py rank %d: the decomposed parity leg failed: %s\n" % (rank, str(ex)[:300]))
        if not all_ok(err is None):
            return None, False, direct_on
        parts = [None] * world if rank == 0 else None
        dist.gather_object({k: mine[k] for k in ("tag", "x", "v", "omega")}, parts, dst=0)
        par, ok = None, True
        if rank == 0:
            one = build_engine(gbed, synthetic.hertz_script(gbed, **kw))
            one.setup()
            one.step(sub)
            ref = one.get_state()
            one_builds = int(one.info().nbuilds)
            del one
            tag = np.concatenate([q["tag"] for q in parts])
            o = np.argsort(tag, kind="stable")
            same = bool(len(tag) == len(ref["tag"]) and np.array_equal(tag[o], ref["tag"]))
            d = float(np.max(gbed["diameter"]))
            L3 = np.array(gbed["boxhi"]) - np.array(gbed["boxlo"])

            def rel(a, b):
                sc = float(np.max(np.abs(b)))
                return float(np.max(np.abs(a - b)) / (sc if sc > 0 else 1.0))
            par = {"against": "the same %d-particle bed on ONE domain (rank 0's GPU), setup + %d sub-steps from the same "
                              "start; the %d domains gathered by tag" % (len(ref["tag"]), sub, world),
                   "n": int(len(ref["tag"])), "substeps": sub, "tags_identical": same,
                   "halo": {0: "RCCL exchange", 1: "direct ghost writes (receive areas + one unpack kernel per exchange)",
                            2: "ghost slots (border records straight into the neighbours' ghost records, no kernel "
                               "between two sub-step kernels)"}.get(direct_on, "direct ghost writes")}
            if same:
                dx = np.concatenate([q["x"] for q in parts])[o] - ref["x"]
                per = np.array(gbed["periodic"], bool)
                dx[:, per] -= L3[per] * np.round(dx[:, per] / L3[per])   # (an atom is wrapped when its owner reneighbours)
                par.update(max_abs_dx_over_d=float(np.max(np.abs(dx)) / d),
                           max_rel_v=rel(np.concatenate([q["v"] for q in parts])[o], ref["v"]),
                           max_rel_omega=rel(np.concatenate([q["omega"] for q in parts])[o], ref["omega"]))
            par["rebuilds_decomposed_rank0"] = mine_builds
            par["rebuilds_single_domain"] = one_builds
            par["tolerance"] = "x 1e-9 d, v / omega 1e-9 of max (SURVEY.md 8d)"
            par["ok"] = bool(same and par["max_abs_dx_over_d"] <= 1e-9 and par["max_rel_v"] <= 1e-9
                             and par["max_rel_omega"] <= 1e-9)
            ok = par["ok"]
        return par, all_ok(ok), direct_on

    # N > 1, before anything is timed: the parity leg, first with the ghost slots (SF_HALO_DIRECT=auto2: the sub-step
    # kernels write their border records straight into the neighbours' ghost records and hand over with flags, nothing
    # between two sub-step kernels), then -- should that transport not come up, run out of time or disagree with the
    # single-domain run on this node -- with the direct ghost writes into receive areas (auto: one unpack kernel per
    # exchange), then over RCCL; the transport that passed also carries the timed runs.  The line says which one it was.
    parity_first, halo_note = None, None
    parity_ok = True
    if world > 1 and not args.no_parity and args.scaling != "weak":
        asked = os.environ.get("SF_HALO_DIRECT")
        # (--one-gpu: ranks that SHARE a GPU must not use the ghost slots at bench sizes -- the gates of N - 1 kernels spin in
        # every wave slot of the chip while the kernel they wait for cannot place its last workgroups; measured: the bounded
        # wait runs out.  On one GPU per rank a kernel only ever waits for OTHER GPUs.)
        ladder = (["auto", "0"] if args.one_gpu else ["auto2", "auto", "0"]) if asked is None else \
            ([asked] if asked == "0" else [asked, "0"])
        if os.environ.get("SF_BENCH_LADDER"):   # (development: the full ladder on ranks that share a GPU, at small sizes)
            ladder = os.environ["SF_BENCH_LADDER"].split(",")
        tried = []
        passed = {}   # transport -> (parity, what sf_slab_direct_halo said, ms per 50 sub-steps)
        for halo_mode in ladder:
            os.environ["SF_HALO_DIRECT"] = halo_mode
            parity_first, ok_all, was_direct = decomposed_parity()
            if ok_all and was_direct:
                passed[halo_mode] = (parity_first, was_direct, trial_ms[0])
                # ghost slots passed: give the receive-area transport its turn too and keep the faster of the two ON THIS
                # NODE (what the in-kernel hand-off and the small write-through stores cost over xGMI is unmeasured)
                if halo_mode == "auto2" and "auto" in ladder and len(passed) == 1:
                    continue
                break
            if passed:   # (the second of two direct transports did not pass: the first one stands)
                tried.append("SF_HALO_DIRECT=%s did not pass after SF_HALO_DIRECT=%s had" % (halo_mode, next(iter(passed))))
                break
            if ok_all and (halo_mode == ladder[-1] or not grid_used[0]):
                break   # (no brick driver -- slabs, or the gloo wire of --one-gpu --: the direct transports do not apply)
            if ok_all:   # ("auto*": the bring-up failed on some rank and the library fell back to RCCL by itself)
                tried.append("SF_HALO_DIRECT=%s did not come up" % halo_mode)
                continue
            tried.append("SF_HALO_DIRECT=%s %s" % (halo_mode, "disagreed with the single-domain run"
                                                   if parity_first is not None else "did not run through"))
        if passed:
            # every rank must take the same decision: rank 0's clock decides
            names = sorted(passed, key=lambda k: passed[k][2] if passed[k][2] is not None else 1e30)
            pick = [names[0]]
            if dist is not None:
                dist.broadcast_object_list(pick, src=0)
            parity_first, was_direct, _ = passed[pick[0]]
            ok_all = True
            os.environ["SF_HALO_DIRECT"] = pick[0]
            if len(passed) > 1:
                tried.append("both direct transports passed: " + ", ".join("SF_HALO_DIRECT=%s %.3f ms per 50 sub-steps"
                                                                            % (k, passed[k][2]) for k in sorted(passed)))
        if tried:
            halo_note = "tried first: " + "; ".join(tried) + ("" if not ok_all else "; SF_HALO_DIRECT=%s carried the run"
                                                               % os.environ["SF_HALO_DIRECT"])
        parity_ok = ok_all

    # N > 1: BASELINE config C4 -- ONE --particles bed split into `world` spatial domains -- is the headline (`value`,
    # scaling "strong"); the weak-scaling run (every rank one --particles slab) is the side object `weak_scaling`
    side = {}
    lmp = None
    if world > 1 and args.scaling in ("both", "strong"):
        # (the headline first: whatever happens to the side measurement afterwards cannot cost it)
        gbed = synthetic.fcc_bed(ncells, seed=12345 + 3, **bed_kw)
        lmp = make_driver("from_global_bed", gbed)
        elapsed, n_total, launches, kernel_ms, info, _info_after = timed_run(lmp)
        N = info.nlocal
        del gbed
    main_exchange_us, main_rebuild_ms = exchange_us[0], rebuild_ms[0]
    if world > 1 and args.scaling == "weak":
        lmp = make_driver("from_bed", bed)
        elapsed, n_total, launches, kernel_ms, info, _info_after = timed_run(lmp)
        main_exchange_us, main_rebuild_ms = exchange_us[0], rebuild_ms[0]
    elif world > 1 and args.scaling == "both":
        try:
            wdrv = make_driver("from_bed", bed)
            el_w, n_w, l_w, k_w, i_w, _ia = timed_run(wdrv)
            side["weak_scaling"] = side_line(el_w, n_w, l_w, k_w, i_w, "every rank owns one %d-particle slab of a channel "
                                             "%d times as long" % (bed["n"], world), "weak")
            del wdrv
        except Exception as ex:   # noqa: BLE001  (a side measurement: the headline above stands)
            side["weak_scaling"] = {"error": str(ex)[:300]}
    exchange_us[0], rebuild_ms[0] = main_exchange_us, main_rebuild_ms
    if world == 1 and args.slab_driver and args.decomposition == "bricks":
        # (development: one brick exchanging with itself, SF_HALO_SELF_COMM=1 -- tests/trace_selfcomm.sh)
        from sedifoam_amd.halo import BrickDriver
        lmp = BrickDriver.from_global_bed(bed, script, dist, 0, 1, (1, 1, 1))
        elapsed, n_total, launches, kernel_ms, info, _info_after = timed_run(lmp)
    elif world == 1 and args.slab_driver:
        lmp = make_driver("from_bed", bed)
        elapsed, n_total, launches, kernel_ms, info, _info_after = timed_run(lmp)
    elif world == 1:
        lmp = build_engine(bed, script)
        elapsed, n_total, launches, kernel_ms, info, _info_after = timed_run(lmp)
    is_strong = world > 1 and args.scaling != "weak"
    k_half = info.npairs_full / 2.0 / max(info.nlocal, 1)

    value = n_total * args.substeps * args.steps / elapsed
    nlabel = ("%.0fM" % (n_total / 1e6)) if abs(n_total / 1e6 - round(n_total / 1e6)) < 0.01 and n_total >= 1e6 else "%d" % int(n_total)
    b_alg = 284.0 + 52.0 * k_half
    mean_kernel_s = (kernel_ms / max(launches, 1)) * 1e-3
    achieved = b_alg * N / mean_kernel_s / 1e9 if launches else 0.0
    info2 = lmp.info()
    out = {
        "metric": "particle-DEM-substeps/sec",
        "value": value,
        "unit": "particle-substeps/s",
        "n_gpus": args.gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        # (one workload for every N of a --scaling both / strong series: the SAME --particles bed, in one domain at N = 1)
        "scaling": "weak" if args.scaling == "weak" else "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": ("%s, periodic x/z, wall y, gravity + fix fdrag, %d DEM sub-steps per step%s"
                         % ("%s-particle loose disordered (fluidised) Hertz-history bed (jitter 0.3 d, spacing 1.1 d)" % nlabel
                            if args.bed == "fluidised" else
                            "%s-particle monodisperse Hertz-history packing (FCC bed, d=1mm, 2%% overlap)" % nlabel,
                            args.substeps,
                            "; the SAME bed split into %d spatial domains (BASELINE config C4)" % world if is_strong else
                            ("; one such slab per GPU of a channel %d times as long" % world if world > 1 else ""))),
            "particles_total": int(n_total),
            "particles_per_gpu": N, "substeps_per_step": args.substeps, "k_half": round(k_half, 3),
            "neighbor_rebuilds_in_run": int(info2.nbuilds - info.nbuilds),
            **({"halo_exchange_us_per_substep": exchange_us[0]} if exchange_us[0] is not None else {}),
            # (host clock around each rebuild inside the timed region, synchronised; not a rocprof trace, whose ~50 small
            # launches per rebuild cost 2-3 us more each)
            **({("decomposed_rebuild_ms_rank0" if (world > 1 or args.slab_driver) else "neighbor_rebuild_ms"): rebuild_ms[0]}
               if rebuild_ms[0] is not None else {}),
            **({"bed_override": bed_kw} if bed_kw else {}),
            "decomposition": (((("%dx%dx%d bricks" % tuple(grid_used[0])) if grid_used[0] else "x-slabs")
                               + (", C++ driver over a stand-in for librccl through host memory (--one-gpu)"
                                  if transport == "rccl" else ", ghost halo over gloo through host memory (--one-gpu)"))
                              if args.one_gpu else
                              (fallback_note[0] or ("%dx%dx%d bricks, C++ driver (sf_brick_init + sf_slab_*), ghosts straight to "
                                                    "the neighbour bricks over RCCL" % tuple(grid_used[0]) if grid_used[0] else
                                                    "x-slabs, C++ driver (sf_slab_*), ghost halo over RCCL"))) if world > 1 else
                             ("single slab through the halo driver" if args.slab_driver else "single domain"),
            **({"transport": ("stand-in for librccl over host memory" if transport == "rccl" else "gloo through host memory")
                if args.one_gpu else ("torch.distributed point-to-point (Python loop)" if fallback_note[0] else
                                      "RCCL point-to-point (ncclSend/ncclRecv groups) from the C++ driver")} if world > 1 else {}),
            **(comm_info[0] or {}),
        },
        **side,
        "roofline": {
            "bound": "hbm", "kernel": "k_substep<hertz>", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "algorithmic_bytes_per_particle_substep": b_alg,
            "mean_kernel_us": mean_kernel_s * 1e6, "launches_timed": int(launches),
            "traffic": None,
        },
    }
    # HBM traffic per launch of the same kernel on the same workload from the committed rocprofv3 PMC passes (separate
    # passes; the summary file holds the calibration of FETCH_SIZE on known byte counts).  The summary carries a hash of
    # the kernel's sources: counters of another kernel are not reported (traffic = null, traffic_stale says why)
    if args.particles == 1000000 and not bed_kw and world == 1:
        from sedifoam_amd.build import kernel_source_hash
        now = kernel_source_hash()
        for name in PMC_SUMMARIES:
            pmc = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(pmc):
                continue
            try:
                doc = json.load(open(pmc))
                if doc.get("kernel_source_sha256_16") != now:
                    out["roofline"]["traffic_stale"] = ("profiles/%s was collected on kernel sources %s, the library is "
                                                        "built from %s" % (name, doc.get("kernel_source_sha256_16"), now))
                    break
                hb = doc["hbm_bytes_per_launch"]
                out["roofline"]["traffic"] = hb["total_calibrated"]
                out["roofline"]["traffic_source"] = ("profiles/%s: rocprofv3 --pmc bytes per launch of this kernel (source "
                                                     "hash %s) on this workload, WRITE_SIZE + calibrated reads (FETCH_SIZE "
                                                     "counts every coalesced stream at 1/2: calibrated on "
                                                     "tests/micro/stream_bench)" % (name, now))
                out["roofline"]["traffic_raw_counters"] = hb["total_raw"]
            except Exception:   # noqa: BLE001
                pass
            break
    # secondary metric of BASELINE.json: coupled CFD-DEM steps/s (drag closure + drag assembly + S sub-steps +
    # cell owner + void-fraction / Ue scatter + Asrc) through the device-resident enhancedCloud, frozen fluid
    if world == 1 and not args.slab_driver and not args.no_coupled:
        from sedifoam_amd import enhancedCloud
        # cells at least 3 d wide (a centre-counted void fraction above 1 makes every closure return inf)
        mesh_n = np.clip(((bed["boxhi"] - bed["boxlo"]) / 3.0e-3).astype(int), 1, 32)
        dx = (bed["boxhi"] - bed["boxlo"]) / mesh_n
        nc = int(np.prod(mesh_n))
        # diffusion smoothing of gamma / Ue / Uf / Asrc with the reference's defaults (createFields.H:126-149:
        # diffusionBandWidth 0.006, diffusionSteps 6), then the same loop with smoothing off
        for key, band in (("coupled_steps_per_s", 0.006), ("coupled_steps_per_s_unsmoothed", 0.0)):
            cloud = enhancedCloud(lmp, bed["boxlo"], dx, mesh_n,
                                  dict(dragModel="ErgunWenYu", subCycles=1, maxPossibleAlpha=0.65,
                                       diffusionBandWidth=band, diffusionSteps=6),
                                  dict(rhob=1000.0, nub=1.0e-6), deltaT=args.substeps * kw["dt"])
            cloud.setFluid(Uf=np.tile([0.0, 0.05, 0.0], (nc, 1)), gradp=np.tile([0.0, -9810.0, 0.0], (nc, 1)))
            cloud.calcTcFields()   # lammpsFoam.C includes liftDragCoeffs.H (alpha cap + calcTcFields) before the time loop
            for _ in range(2):
                cloud.evolve(); cloud.calcTcFields()
            barrier()
            t1 = time.perf_counter()
            ncpl = max(3, args.steps // 2)
            for _ in range(ncpl):
                cloud.evolve(); cloud.calcTcFields()
            barrier()
            out["config"][key] = ncpl / (time.perf_counter() - t1)
            if band:
                # the reference's timer buckets (writeCPUTime.H:1-19), per coupled step; the same names are roctx ranges
                tm = cloud.cpuTimeSplit()
                nrun = ncpl + 3
                out["config"]["coupled_buckets_ms"] = {
                    "evolve": 1e3 * tm["evolve"] / nrun, "calcTcField": 1e3 * tm["calcTc"] / nrun,
                    "foam->lammps (drag closure + assembly)": 1e3 * tm["dragOnParticles"] / nrun,
                    "lammps (%d sub-steps)" % args.substeps: 1e3 * tm["lammps"] / nrun,
                    "particle move (cell owner + scatter + smoothing)": 1e3 * tm["scatter"] / nrun}
            cloud.close()
        out["config"]["coupled_step"] = ("ErgunWenYu drag + %d DEM sub-steps + scatter + Asrc + diffusion smoothing "
                                         "(b = 6 mm, 6 steps), %dx%dx%d mesh" % (args.substeps, mesh_n[0], mesh_n[1],
                                                                                  mesh_n[2]))
        # the library.h drop-in boundary (softParticleCloud.C:838-922): per CFD step the caller hands over HOST
        # arrays (lammps_put_local_info), runs the sub-steps and reads HOST arrays back (lammps_get_local_info);
        # this rate includes the PCIe copies and the by-tag scatter, it is reported next to `value`, never as it
        n_loc = lmp.get_local_n()
        st = lmp.get_local_info()
        fd = np.zeros((n_loc, 3)); tags = np.ascontiguousarray(st["tag"])
        lmp.put_local_info(fd, tags, DuDt=np.zeros((n_loc, 3)), foamCpuId=np.zeros(n_loc, np.int32))
        lmp.step(args.substeps); lmp.get_local_info()
        barrier()
        t2 = time.perf_counter()
        nlib = max(3, args.steps // 2)
        for _ in range(nlib):
            lmp.put_local_info(fd, tags, DuDt=np.zeros((n_loc, 3)), foamCpuId=np.zeros(n_loc, np.int32))
            lmp.step(args.substeps)
            st = lmp.get_local_info()
        barrier()
        out["config"]["host_boundary_substeps_per_s"] = n_loc * args.substeps * nlib / (time.perf_counter() - t2)
        out["config"]["host_boundary"] = ("lammps_put_local_info + lammps_step(%d) + lammps_get_local_info on host "
                                          "arrays (PCIe-inclusive)" % args.substeps)
    if (world > 1 or args.slab_driver) and args.coupled_multi:
        from sedifoam_amd import enhancedCloud
        glo = np.array(bed["boxlo"], dtype=np.float64)
        ghi = np.array(bed["boxhi"], dtype=np.float64)
        mesh_n = np.clip(((bed["boxhi"] - bed["boxlo"]) / 3.0e-3).astype(int), 1, 32)
        if is_strong:     # the bed's own box, cell layers along x a multiple of the ranks
            mesh_n[0] = max(world, (int(mesh_n[0]) // world) * world)
        else:             # one bed per rank, side by side
            ghi[0] = glo[0] + world * (bed["boxhi"][0] - bed["boxlo"][0])
            mesh_n[0] *= world
        dx = (ghi - glo) / mesh_n
        nc = int(np.prod(mesh_n))
        # the mesh cut by the slab planes (every rank its nx / N layers + ghost layers, exchanges in C++ over the engine's
        # communicator) whenever the particles are in x-slabs and the layers divide; else the whole mesh on every rank
        part = grid_used[0] is None and int(mesh_n[0]) % world == 0 and not args.coupled_replicated
        cloud = enhancedCloud(lmp.e.lmp, glo, dx, mesh_n,
                              dict(dragModel="ErgunWenYu", subCycles=1, maxPossibleAlpha=0.65,
                                   diffusionBandWidth=0.006, diffusionSteps=6),
                              dict(rhob=1000.0, nub=1.0e-6), deltaT=args.substeps * kw["dt"], driver=lmp,
                              mesh_partition=part, mesh_periodic=(1, 0, 1) if part else None)
        cloud.setFluid(Uf=np.tile([0.0, 0.05, 0.0], (nc, 1)), gradp=np.tile([0.0, -9810.0, 0.0], (nc, 1)))
        cloud.calcTcFields()
        cloud.evolve(); cloud.calcTcFields()
        barrier()
        t1 = time.perf_counter()
        ncpl = max(3, args.steps // 2)
        for _ in range(ncpl):
            cloud.evolve(); cloud.calcTcFields()
        barrier()
        out["config"]["coupled_steps_per_s"] = ncpl / (time.perf_counter() - t1)
        out["config"]["coupled_step"] = ("decomposed particles, %dx%dx%d mesh %s; ErgunWenYu + %d sub-steps + scatter + Asrc "
                                         "+ smoothing (6 mm, 6 steps)"
                                         % (mesh_n[0], mesh_n[1], mesh_n[2],
                                            "cut by the slab planes (face-halo adds + x-line all-to-all in C++ over RCCL)"
                                            if part else "on every rank, all-reduced per-cell sums", args.substeps))
    # N = 1: the loose disordered ("fluidised") bed of the same size next to the headline -- BASELINE config C3 is a
    # fluidised bed; the lattice of the headline is the best case (every listed neighbour touches, a rebuild every ~170
    # sub-steps).  Same engine path, same timing rules, shorter run.
    if world == 1 and not args.slab_driver and not args.no_fluidised and args.bed == "packed" and not bed_kw:
        fbed = synthetic.fcc_bed(ncells, seed=12345 + 3, **FLUIDISED)
        flmp = build_engine(fbed, synthetic.hertz_script(fbed, **kw))
        keep = (args.steps, args.warmup)
        args.steps, args.warmup = max(2, args.steps), 1   # (as many steps as the headline: 0.12 s of GPU time)
        el_f, n_f, l_f, k_f, i_f, i_f1 = timed_run(flmp)
        kh_f = i_f1.npairs_full / 2.0 / max(i_f1.nlocal, 1)
        fo = {"value": n_f * args.substeps * args.steps / el_f, "unit": "particle-substeps/s",
              "ms_per_step": 1e3 * el_f / args.steps, "steps": args.steps, "warmup": args.warmup,
              "workload": "%d-particle loose disordered Hertz bed (FCC sites at spacing 1.1 d, jitter 0.3 d), same fixes, %d "
                          "sub-steps per step" % (int(n_f), args.substeps),
              "k_half": round(kh_f, 3), "neighbor_rebuilds_in_run": int(i_f1.nbuilds - i_f.nbuilds)}
        if l_f:
            fo["mean_kernel_us"] = 1e3 * k_f / l_f
            fo["roofline_frac"] = (284.0 + 52.0 * kh_f) * i_f1.nlocal / (1e-3 * k_f / l_f) / 1e9 / HBM_PEAK_GBS
            fo["roofline_frac_whole_run"] = (284.0 + 52.0 * kh_f) * fo["value"] / 1e9 / HBM_PEAK_GBS
        if rebuild_ms[0] is not None:
            fo["neighbor_rebuild_ms"] = rebuild_ms[0]
        out["fluidised_bed"] = fo
        args.steps, args.warmup = keep
        del flmp, fbed
    # N = 1: the kernel's roofline fraction on a 2 M-grain bed of the same packing next to the 1 M one (the 1 M launch is
    # five rounds of resident waves: its fill and drain weigh ~9 % of it, half that at 2 M), and BASELINE.json's other
    # single-GPU configurations -- C2 (10 k grains, 32^3 mesh), C3 (100 k fluidised), C5 (500 k polydisperse + cohesive +
    # lubricate/poly) -- each through the same engine path and timing rules, with its coupled step where it has a mesh
    cfg_cases = None
    if (world == 1 and not args.slab_driver and not args.no_configs and args.bed == "packed" and not bed_kw
            and args.particles == 1000000):
        keep = (args.steps, args.warmup)
        b2 = synthetic.fcc_bed(synthetic.fcc_cells_for(2000000), seed=12345 + 3)
        l2 = build_engine(b2, synthetic.hertz_script(b2, **kw))
        args.steps, args.warmup = max(2, keep[0] // 3), 1
        el2, n2, la2, km2, i2, i2b = timed_run(l2)
        if la2:
            kh2 = i2b.npairs_full / 2.0 / max(i2b.nlocal, 1)
            out["roofline"]["frac_2m"] = (284.0 + 52.0 * kh2) * i2b.nlocal / (1e-3 * km2 / la2) / 1e9 / HBM_PEAK_GBS
            out["roofline"]["mean_kernel_us_2m"] = 1e3 * km2 / la2
            out["roofline"]["particles_2m"] = int(n2)
            out["roofline"]["value_2m"] = n2 * args.substeps * args.steps / el2
        del l2, b2
        cfg_cases = config_cases(synthetic)
        out["configs"] = {}
        for name, (cbed, ccfg, cmesh, label) in cfg_cases.items():
            try:
                clmp = build_engine(cbed, config_script(cbed, ccfg))
                # (C2 / C3: tens of milliseconds per step -- twice the headline's steps; C5_wide: bounded by what its
                # lubrication series lets the bed live, C5W_SUBSTEPS)
                args.steps, args.warmup = {"C2": (2 * keep[0], 2), "C3": (2 * keep[0], 2), "C5": (max(2, keep[0] // 2), 1),
                                            "C5_wide": (max(2, keep[0] // 3), 1)}[name]
                sub_keep = args.substeps
                if name == "C5_wide":
                    args.substeps = C5W_SUBSTEPS
                try:
                    el_c, n_c, l_c, k_c, i_c, i_c1 = timed_run(clmp)
                    sub_c = args.substeps
                finally:
                    args.substeps = sub_keep
                kh_c = i_c1.npairs_full / 2.0 / max(i_c1.nlocal, 1)
                o = {"workload": label, "value": n_c * sub_c * args.steps / el_c, "unit": "particle-substeps/s",
                     "ms_per_step": 1e3 * el_c / args.steps, "steps": args.steps, "warmup": args.warmup, "substeps": sub_c,
                     "particles": int(n_c), "k_half": round(kh_c, 3), "longest_row": int(i_c1.max_neigh_used),
                     "neighbor_rebuilds_in_run": int(i_c1.nbuilds - i_c.nbuilds),
                     "algorithmic_bytes_per_particle_substep": 284.0 + 52.0 * kh_c}
                if l_c:
                    o["mean_kernel_us"] = 1e3 * k_c / l_c
                    o["roofline_frac"] = (284.0 + 52.0 * kh_c) * i_c1.nlocal / (1e-3 * k_c / l_c) / 1e9 / HBM_PEAK_GBS
                    o["roofline_frac_whole_run"] = (284.0 + 52.0 * kh_c) * o["value"] / 1e9 / HBM_PEAK_GBS
                if cmesh is not None and not args.no_coupled:
                    from sedifoam_amd import enhancedCloud
                    mesh_n = np.array(cmesh, np.int32)
                    dxm = (cbed["boxhi"] - cbed["boxlo"]) / mesh_n
                    ncm = int(np.prod(mesh_n))
                    cloud = enhancedCloud(clmp, cbed["boxlo"], dxm, mesh_n,
                                          dict(dragModel="ErgunWenYu", subCycles=1, maxPossibleAlpha=0.65,
                                               diffusionBandWidth=0.006, diffusionSteps=6),
                                          dict(rhob=1000.0, nub=1.0e-6), deltaT=args.substeps * kw["dt"])
                    cloud.setFluid(Uf=np.tile([0.0, 0.05, 0.0], (ncm, 1)), gradp=np.tile([0.0, -9810.0, 0.0], (ncm, 1)))
                    cloud.calcTcFields()
                    cloud.evolve(); cloud.calcTcFields()
                    barrier()
                    t1 = time.perf_counter()
                    ncpl = max(3, args.steps)
                    for _ in range(ncpl):
                        cloud.evolve(); cloud.calcTcFields()
                    barrier()
                    o["coupled_steps_per_s"] = ncpl / (time.perf_counter() - t1)
                    o["coupled_step"] = ("ErgunWenYu drag + %d DEM sub-steps + scatter + Asrc + diffusion smoothing (b = 6 mm, "
                                         "6 steps), %dx%dx%d mesh" % ((args.substeps,) + tuple(int(k) for k in mesh_n)))
                    cloud.close()
                out["configs"][name] = o
                del clmp
            except Exception as ex:   # noqa: BLE001  (a side measurement: the headline stands)
                out["configs"][name] = {"workload": label, "error": str(ex)[:300]}
        args.steps, args.warmup = keep
    if world > 1 and rank == 0:
        if parity_first is not None:
            out["parity"] = parity_first
        try:
            direct_on = int(lmp.e.lmp.L.sf_slab_direct_halo(lmp.e.lmp.ptr)) if grid_used[0] else 0
        except Exception:   # noqa: BLE001
            direct_on = 0
        out["config"]["halo"] = {
            2: "ghost slots: the sub-step kernel writes the border records straight into the ghost range of the neighbours' "
               "record arrays, its last wave publishes flag | vote, the next sub-step kernel waits at its gate -- no kernel "
               "between two sub-step kernels",
            1: "direct ghost writes: the sub-step kernel writes the border records into the neighbours' IPC-mapped receive "
               "areas, one kernel per exchange publishes / awaits the ranks' flags",
        }.get(direct_on, "one grouped ncclSend/ncclRecv per sub-step + unpack kernel")
        if halo_note:
            out["config"]["halo_note"] = halo_note
    if rank == 0 and not args.no_cpu_baseline:   # (on rank 0's host cores; at N > 1 the other ranks are done)
        sample_n = args.cpu_sample or 1000000
        sub = 50
        do_parity = (world == 1 and not args.no_parity and not args.slab_driver and sample_n == args.particles and not bed_kw)
        gpu_state = None
        if do_parity:
            # the SAME bed from the SAME start through the product path: setup + `sub` sub-steps, outside the timed region
            plmp = build_engine(bed, script)
            plmp.setup()
            plmp.step(sub)
            gpu_state = plmp.get_state()
            p_builds = int(plmp.info().nbuilds)
            del plmp
        res = cpu_baseline(synthetic.fcc_cells_for(sample_n), kw, sub, keep_state=True)
        v, n_s, secs = res[:3]
        out["cpu_baseline"] = {"value": v, "unit": "particle-substeps/s", "cores": 1, "kind": "port",
                               "buckets_ms_per_step": {"lammps (%d sub-steps)" % sub: 1e3 * secs},
                               "sample": "%d-particle bed of the same packing, %d sub-steps, %.1f s, "
                                         "oracle/ (C, gcc -O2) single thread" % (n_s, sub, secs)}
        if not args.no_coupled:
            # BASELINE.json's second metric on the CPU: ONE coupled CFD-DEM step of the same kind the GPU's
            # `coupled_steps_per_s` times, continued from the oracle state above, in the reference's timer buckets
            cbed = res[5]
            cmesh = np.clip(((cbed["boxhi"] - cbed["boxlo"]) / 3.0e-3).astype(int), 1, 32)
            csecs, cb = cpu_coupled_step(res[4], cbed, cmesh, sub, 0.006, 6)
            out["cpu_baseline"]["coupled_steps_per_s"] = 1.0 / csecs
            out["cpu_baseline"]["coupled_buckets_ms"] = cb
            out["cpu_baseline"]["coupled_sample"] = ("one coupled step (ErgunWenYu drag + %d DEM sub-steps + scatter + Asrc + "
                                                     "diffusion smoothing b = 6 mm, 6 steps; %dx%dx%d mesh) of the same %d-"
                                                     "particle bed, %.1f s, oracle/ single thread"
                                                     % (sub, cmesh[0], cmesh[1], cmesh[2], n_s, csecs))
        if do_parity:
            o = res[3]
            d = float(np.max(bed["diameter"]))
            same = bool(np.array_equal(o["tag"], gpu_state["tag"]))

            def rel(a, b):
                sc = float(np.max(np.abs(b)))
                return float(np.max(np.abs(a - b)) / (sc if sc > 0 else 1.0))
            out["parity"] = {"against": "oracle/ (CPU restatement of the reference) from the same start, setup + %d sub-steps, "
                                        "the whole %d-particle bed of the headline" % (sub, n_s),
                             "n": int(n_s), "substeps": sub, "tags_identical": same,
                             "max_abs_dx_over_d": float(np.max(np.abs(gpu_state["x"] - o["x"])) / d) if same else None,
                             "max_rel_v": rel(gpu_state["v"], o["v"]) if same else None,
                             "max_rel_omega": rel(gpu_state["omega"], o["omega"]) if same else None,
                             "max_rel_f": rel(gpu_state["f"], o["f"]) if same else None,
                             "tolerance": "x 1e-9 d, v / omega 1e-9 of max (SURVEY.md 8d)",
                             "gpu_rebuilds": p_builds}
            out["parity"]["ok"] = bool(same and out["parity"]["max_abs_dx_over_d"] <= 1e-9
                                       and out["parity"]["max_rel_v"] <= 1e-9 and out["parity"]["max_rel_omega"] <= 1e-9)
            parity_ok = out["parity"]["ok"]
        del res
        if cfg_cases is not None:
            # the named configurations on the CPU: the oracle on the same bed and script, a bounded number of sub-steps
            # (200 / 50 / 10 for 10 k / 100 k / 500 k grains: 0.5-20 s each)
            for name, (cbed, ccfg, cmesh, label) in cfg_cases.items():
                if name in out.get("configs", {}):
                    try:
                        out["configs"][name]["cpu_baseline"] = config_cpu_leg(cbed, ccfg, {"C2": 200, "C3": 50}.get(name, 10))
                    except Exception as ex:   # noqa: BLE001
                        out["configs"][name]["cpu_baseline"] = {"error": str(ex)[:300]}
            cfg_cases = None
        allc = cpu_baseline_all_cores(args.cpu_all_particles, 20) if args.cpu_all_particles > 0 else None
        if allc:
            out["cpu_baseline_all_cores"] = allc
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0 and not parity_ok:
        sys.stderr.write("bench.py: the parity leg FAILED (see the `parity` object of the line)\n")
        raise SystemExit(1)


if __name__ == "__main__":
    def _arg(name, default):
        a = sys.argv[1:]
        return int(a[a.index(name) + 1]) if name in a and a.index(name) + 1 < len(a) else default
    try:
        main()
    except SystemExit as ex:
        # (SystemExit with a message = the run refused to start -- no GPU, wrong launcher: say so in the line as well)
        if isinstance(ex.code, str) and int(os.environ.get("RANK", "0")) == 0 and "--cpu-worker" not in sys.argv:
            print(error_line(_arg("--gpus", 1), _arg("--steps", 10), _arg("--warmup", 2), ex.code))
            sys.stdout.flush()
        raise
    except BaseException as ex:   # noqa: BLE001  -- a run that dies before its line still prints ONE parseable line (rank 0) ...
        import traceback
        traceback.print_exc()
        if int(os.environ.get("RANK", "0")) == 0:
            print(error_line(_arg("--gpus", 1), _arg("--steps", 10), _arg("--warmup", 2), "%s: %s" % (type(ex).__name__, ex)))
            sys.stdout.flush()
        os._exit(1)   # ... and does not wait in a destructor for ranks that are stuck in a collective
