#!/usr/bin/env python3
"""bench.py -- particle-DEM-substeps/s of the sediFoam hot path on MI355X.

One "step" = one `lammps_step(S)` call (S = 50 DEM sub-steps, BASELINE.json configs[2..3]) of the fused
Hertz-history contact / fix fdrag / wall / gravity / nve-sphere kernel over a synthetic 1 M-particle
monodisperse Hertz packing, particle state resident in HBM before timing starts.
With --gpus N (launched through torch.distributed.run, one rank per GPU) every rank owns one such slab
of a periodic channel N times as long, with a ghost-particle halo over RCCL: weak scaling.

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
PMC_SUMMARY = "r02_d_pmc_summary.json"   # the committed PMC passes of the current kernel (tests/profile_round.sh)


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


def cpu_baseline(ncells, script_kw, substeps, seed=12345 + 3):
    """Oracle (CPU port of the reference algorithm) on the same kind of bed: particle-substeps/s."""
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
    return bed["n"] * substeps / dt, bed["n"], dt


KW = dict(kn=1.0e7, gamman=0.5, xmu=0.4, dt=1.0e-6, skin_d=0.25, g=9.81)


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
                    help="N > 1: weak = every rank owns one --particles slab (the `value` of the JSON line); strong = "
                         "--particles in total, split into N x-slabs (BASELINE config C4), reported as "
                         "`strong_scaling` next to it; both (default) measures one after the other.  N = 1: identical")
    ap.add_argument("--one-gpu", action="store_true",
                    help="development: all ranks share GPU 0, halo over gloo through host memory (RCCL refuses two "
                         "ranks on one device); exercises the N > 1 code on a 1-GPU box")
    ap.add_argument("--coupled-multi", action="store_true",
                    help="with --gpus N > 1 also time coupled steps: enhancedCloud over the decomposed particles, "
                         "whole mesh on every rank, per-cell sums all-reduced")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, args.gpus))

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    if args.one_gpu:
        local_rank = 0
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
        if args.one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from sedifoam_amd import synthetic
    kw = KW
    ncells = synthetic.fcc_cells_for(args.particles)
    bed_kw = {}
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
        lmp.setup()
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

    def make_driver(factory, the_bed):
        """the C++ driver over RCCL; if it cannot come up on EVERY rank alike (an exception, not a hang), say so loudly
        and measure the same protocol driven from Python over torch.distributed instead of measuring nothing"""
        from sedifoam_amd.halo import SlabDriver
        err = None
        try:
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
            return drv
        if args.one_gpu or transport not in (None, "rccl"):
            raise err if err is not None else RuntimeError("another rank could not create its halo driver")
        sys.stderr.write("bench.py: the C++ RCCL halo driver did not come up (%s) -- FALLING BACK to the Python loop over "
                         "torch.distributed (slower; config.decomposition says so)\n" % (err,))
        fallback_note[0] = "x-slabs, ghost halo driven from Python over torch.distributed (the C++ RCCL driver failed: %s)" % (err,)
        return getattr(SlabDriver, factory)(the_bed, script, dist, rank, world, transport="direct")

    strong = None
    if world > 1 and args.scaling in ("both", "strong"):
        # BASELINE config C4: ONE --particles bed, split into `world` x-slabs (strong scaling)
        from sedifoam_amd.halo import SlabDriver
        gbed = synthetic.fcc_bed(ncells, seed=12345 + 3, **bed_kw)
        sdrv = make_driver("from_global_bed", gbed)
        el_s, n_s, _l, _k, _i0, _i1 = timed_run(sdrv)
        strong = {"value": n_s * args.substeps * args.steps / el_s, "unit": "particle-substeps/s",
                  "halo_exchange_us_per_substep": exchange_us[0],
                  "ms_per_step": 1e3 * el_s / args.steps, "particles_total": int(n_s), "scaling": "strong",
                  "workload": "the SAME %d-particle bed split into %d x-slabs (BASELINE config C4)" % (int(n_s), world)}
        if args.scaling == "strong":
            elapsed, n_total, launches, kernel_ms, info = el_s, n_s, _l, _k, _i0
            lmp = sdrv
            N = info.nlocal
        else:
            del sdrv
        del gbed

    if world > 1 and args.scaling == "strong":
        pass
    elif world > 1 or args.slab_driver:
        from sedifoam_amd.halo import SlabDriver
        lmp = make_driver("from_bed", bed)
        elapsed, n_total, launches, kernel_ms, info, _info_after = timed_run(lmp)
    else:
        lmp = build_engine(bed, script)
        elapsed, n_total, launches, kernel_ms, info, _info_after = timed_run(lmp)
    k_half = info.npairs_full / 2.0 / max(info.nlocal, 1)

    value = n_total * args.substeps * args.steps / elapsed
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
        "scaling": "strong" if (world > 1 and args.scaling == "strong") else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "1M-particle monodisperse Hertz-history packing (FCC bed, d=1mm, 2%% overlap), "
                        "periodic x/z, wall y, gravity + fix fdrag, %d DEM sub-steps per step" % args.substeps,
            "particles_per_gpu": N, "substeps_per_step": args.substeps, "k_half": round(k_half, 3),
            "neighbor_rebuilds_in_run": int(info2.nbuilds - info.nbuilds),
            **({"halo_exchange_us_per_substep": exchange_us[0]} if exchange_us[0] is not None else {}),
            **({"decomposed_rebuild_ms_rank0": rebuild_ms[0]} if rebuild_ms[0] is not None else {}),
            **({"bed_override": bed_kw} if bed_kw else {}),
            "decomposition": (("x-slabs, C++ driver over a stand-in for librccl through host memory (--one-gpu)"
                               if transport == "rccl" else
                               "x-slabs, ghost halo over gloo through host memory (--one-gpu)") if args.one_gpu else
                              (fallback_note[0] or "x-slabs, C++ driver (sf_slab_*), ghost halo over RCCL")) if world > 1 else
                             ("single slab through the halo driver" if args.slab_driver else "single domain"),
        },
        **({"strong_scaling": strong} if strong else {}),
        "roofline": {
            "bound": "hbm", "kernel": "k_substep<hertz>", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "algorithmic_bytes_per_particle_substep": b_alg,
            "mean_kernel_us": mean_kernel_s * 1e6, "launches_timed": int(launches),
            "traffic": None,
        },
    }
    # HBM traffic per launch of the same kernel on the same workload from the committed rocprofv3 PMC passes
    # (separate passes; the summary file holds the calibration of FETCH_SIZE on known byte counts)
    pmc = os.path.join(ROOT, "profiles", PMC_SUMMARY)
    if os.path.exists(pmc) and args.particles == 1000000 and not bed_kw:
        try:
            hb = json.load(open(pmc))["hbm_bytes_per_launch"]
            out["roofline"]["traffic"] = hb["total_calibrated"]
            out["roofline"]["traffic_source"] = ("profiles/%s: rocprofv3 --pmc bytes per launch of this kernel on this "
                                                 "workload, WRITE_SIZE + calibrated reads (FETCH_SIZE counts every "
                                                 "coalesced stream at 1/2: calibrated on tests/micro/stream_bench)"
                                                 % PMC_SUMMARY)
            out["roofline"]["traffic_raw_counters"] = hb["total_raw"]
        except Exception:
            pass
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
        ghi[0] = glo[0] + world * (bed["boxhi"][0] - bed["boxlo"][0])
        mesh_n = np.clip(((bed["boxhi"] - bed["boxlo"]) / 3.0e-3).astype(int), 1, 32)
        mesh_n[0] *= world
        dx = (ghi - glo) / mesh_n
        nc = int(np.prod(mesh_n))
        cloud = enhancedCloud(lmp.e.lmp, glo, dx, mesh_n,
                              dict(dragModel="ErgunWenYu", subCycles=1, maxPossibleAlpha=0.65,
                                   diffusionBandWidth=0.006, diffusionSteps=6),
                              dict(rhob=1000.0, nub=1.0e-6), deltaT=args.substeps * kw["dt"], driver=lmp)
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
        out["config"]["coupled_step"] = ("decomposed particles, %dx%dx%d mesh on every rank, all-reduced per-cell sums; "
                                         "ErgunWenYu + %d sub-steps + scatter + Asrc + smoothing (6 mm, 6 steps)"
                                         % (mesh_n[0], mesh_n[1], mesh_n[2], args.substeps))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # (the CPU baseline is an N = 1 measurement)
        sample_n = args.cpu_sample or 1000000
        sub = 50
        v, n_s, secs = cpu_baseline(synthetic.fcc_cells_for(sample_n), kw, sub)
        out["cpu_baseline"] = {"value": v, "unit": "particle-substeps/s", "cores": 1, "kind": "port",
                               "buckets_ms_per_step": {"lammps (%d sub-steps)" % sub: 1e3 * secs},
                               "sample": "%d-particle bed of the same packing, %d sub-steps, %.1f s, "
                                         "oracle/ (C, gcc -O2) single thread" % (n_s, sub, secs)}
        allc = cpu_baseline_all_cores(args.cpu_all_particles, 20) if args.cpu_all_particles > 0 else None
        if allc:
            out["cpu_baseline_all_cores"] = allc
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
