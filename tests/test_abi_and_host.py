"""CPU-side checks (no GPU, no compute): the C-ABI library loads and exports every symbol that
include/sedifoam_amd.h declares, host-only entry points behave like the reference, and the product refuses to
run without a HIP device instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sedifoam_amd.h")


def _declared_symbols():
    txt = open(HEADER).read()
    txt = txt.split("#ifdef SEDIFOAM_AMD_LAMMPS_NAMES")[0]          # the inline aliases are not exports
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sfk?_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import sedifoam_amd
    L = C.CDLL(sedifoam_amd._lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 60
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    # and the python binding table covers the same set
    assert set(sedifoam_amd.exported_symbols()) == set(names)
    sedifoam_amd.lib()


def test_header_cites_reference_interface():
    txt = open(HEADER).read()
    for needle in ("library.h:29", "library.h:58", "library.cpp:344-366", "pair_gran_hertzFix_history.cpp:45-287",
                   "fix_cohesive.cpp:138-263", "pair_lubricate_poly.cpp:65-444", "fix_fluid_drag.cpp:114-164",
                   "ErgunWenYu.C:86-145", "enhancedCloud.C:669-787", "enhancedCloud.C:316-441",
                   "softParticleCloud.C:209-261"):
        assert needle in txt, needle


def _gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_gpu(), reason="checks the no-device behaviour")
def test_no_device_means_loud_failure_not_fallback():
    import sedifoam_amd
    L = sedifoam_amd.lib()
    assert L.sf_device_check() != 0
    with pytest.raises(sedifoam_amd.SfError, match="no HIP device"):
        sedifoam_amd.Lammps()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sedifoam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "from oracle" not in src and "import oracle" not in src and "sedifoam_oracle" not in src, f


def test_adjust_lamp_timestep_host_logic():
    # softParticleCloud.C:209-261 : xiaocase3 (2e-5 / 2e-7, 1 sub-cycle), multiParticles (1e-3 / 1e-5, 2)
    from sedifoam_amd import adjustLampTimestep, SfError
    r = adjustLampTimestep(2e-5, 2e-7, 1)
    assert r == dict(dtLampAdj=pytest.approx(2e-7), solidStepsPerDt=100, subCycles=1, subSteps=100)
    r = adjustLampTimestep(1e-3, 1e-5, 2)
    assert (r["solidStepsPerDt"], r["subCycles"], r["subSteps"]) == (100, 2, 50)
    r = adjustLampTimestep(1e-3, 3e-5, 4)          # round(33.3) = 33 -> (33/4)*4 = 32 steps, dt = 1e-3/33
    assert r["solidStepsPerDt"] == 32 and r["subSteps"] == 8 and r["dtLampAdj"] == pytest.approx(1e-3 / 33)
    r = adjustLampTimestep(1e-6, 1e-5, 3)          # dnSub = round(0.1) = 0 -> 1 ; subCycles clipped to 0 steps
    assert r["subSteps"] == 1
    # and it agrees with the oracle's restatement
    from oracle import binding as ob
    dt = C.c_double(); st = C.c_int(); sc = C.c_int(); ss = C.c_int()
    for args in ((2e-5, 2e-7, 1), (1e-3, 1e-5, 2), (1e-3, 3e-5, 4), (5e-4, 7e-6, 5)):
        rc = ob.lib().orc_adjust_timestep(*args, C.byref(dt), C.byref(st), C.byref(sc), C.byref(ss))
        if rc == 0:
            r = adjustLampTimestep(*args)
            assert (r["dtLampAdj"], r["solidStepsPerDt"], r["subCycles"], r["subSteps"]) == (
                dt.value, st.value, sc.value, ss.value)
        else:
            with pytest.raises(SfError):
                adjustLampTimestep(*args)


def test_gran_settings_like_pair_style_parser():
    # pair_gran_hertzFix_history.cpp:293-317
    import sedifoam_amd
    from sedifoam_amd._lib import GranParams
    L = sedifoam_amd.lib()
    p = GranParams()
    assert L.sfk_gran_settings(C.byref(p), 1e7, 1, 0.0, 0.5, 1, 0.0, 0.4, 1, 1.0) == 0
    assert p.kt == pytest.approx(1e7 * 2.0 / 7.0) and p.gammat == 0.25
    assert L.sfk_gran_settings(C.byref(p), 1e7, 1, 0.0, 0.5, 1, 0.0, 0.4, 0, 1.0) == 0 and p.gammat == 0.0
    assert L.sfk_gran_settings(C.byref(p), -1.0, 1, 0.0, 0.5, 1, 0.0, 0.4, 1, 1.0) == -1
    assert b"Illegal pair_style" in L.sf_last_error()
    assert L.sfk_gran_settings(C.byref(p), 1.0, 1, 0.0, 0.5, 1, 0.0, 0.4, 2, 1.0) == -1


def test_bench_cpu_legs_run_without_a_gpu():
    """bench.py's CPU baseline legs (the oracle timed on host cores) are plain host code: the single-core leg and the
    one-process-per-usable-core leg both run here, on tiny samples."""
    import bench
    v, n, secs = bench.cpu_baseline((3, 3, 3), bench.KW, 2)
    assert n == 108 and v > 0 and secs > 0
    cores = bench._usable_cores()
    assert 1 <= cores <= len(__import__("os").sched_getaffinity(0))
    r = bench.cpu_baseline_all_cores(500, 2)
    assert r["cores"] == cores and r["value"] > 0 and r["kind"] == "port"


@pytest.mark.parametrize("grid,periodic", [((2, 2, 2), (1, 1, 1)), ((2, 2, 2), (1, 0, 1)), ((4, 1, 2), (1, 0, 1)),
                                           ((3, 2, 1), (1, 1, 0)), ((2, 1, 1), (1, 0, 1)), ((3, 3, 3), (0, 1, 1)),
                                           ((1, 1, 4), (1, 0, 1))])
def test_brick_exchange_pattern_is_consistent_between_every_pair_of_ranks(grid, periodic):
    """The host logic of the brick driver (sf_brick_init) without a device: for every pair of ranks, the blocks the
    sender lists for the receiver -- in its send order -- are the blocks the receiver expects from the sender, in its
    receive order (one grouped ncclSend / ncclRecv per sub-step relies on it); directions never point through a
    non-periodic box face or along a dimension the grid does not cut; the face neighbours of the staged migration
    are mutual."""
    import ctypes as C
    import sedifoam_amd
    L = sedifoam_amd.lib()
    world = grid[0] * grid[1] * grid[2]
    per = (C.c_int * 3)(*periodic)
    pat = []
    for r in range(world):
        ns, nr = C.c_int(), C.c_int()
        sp, sc, rp, rc = ((C.c_int * 26)() for _ in range(4))
        fn = (C.c_int * 6)()
        assert L.sf_brick_pattern(r, grid[0], grid[1], grid[2], per, C.byref(ns), sp, sc, C.byref(nr), rp, rc, fn) == 0
        pat.append(dict(send=[(sp[q], sc[q]) for q in range(ns.value)], recv=[(rp[q], rc[q]) for q in range(nr.value)],
                        face=list(fn)))
    coord = lambda r: (r % grid[0], (r // grid[0]) % grid[1], r // (grid[0] * grid[1]))
    for a in range(world):
        assert pat[a]["send"] == sorted(pat[a]["send"]) and pat[a]["recv"] == sorted(pat[a]["recv"])
        for b in range(world):
            to_b = [c for p, c in pat[a]["send"] if p == b]
            from_a = [c for p, c in pat[b]["recv"] if p == a]
            assert to_b == from_a, (a, b)
        ca = coord(a)
        for peer, code in pat[a]["send"]:
            d = (code % 3 - 1, (code // 3) % 3 - 1, code // 9 - 1)
            want = []
            for k in range(3):
                assert d[k] == 0 or grid[k] > 1                  # uncut dimensions keep their images local
                n = ca[k] + d[k]
                assert 0 <= n < grid[k] or periodic[k]           # never through a wall
                want.append(n % grid[k])
            assert peer == want[0] + grid[0] * (want[1] + grid[1] * want[2])
        for k in range(3):
            for side in range(2):
                nb = pat[a]["face"][2 * k + side]
                if nb >= 0:
                    assert pat[nb]["face"][2 * k + (1 - side)] == a
    # a fully periodic 2 x 2 x 2 grid: every rank exchanges with all 7 others (26 blocks)
    if grid == (2, 2, 2) and periodic == (1, 1, 1):
        assert all(len(p["send"]) == 26 and len({q for q, _ in p["send"]}) == 7 for p in pat)


def test_processors_line_resolves_like_lammps():
    """`processors px py pz` with `*` entries: [3P] ProcMap::onelevel_grid's rule (least sub-domain surface, px slowest,
    first of equals) -- host logic of the -parallel bring-up, no device needed"""
    import ctypes as C
    import sedifoam_amd
    L = sedifoam_amd.lib()

    def grid(world, box, user=(0, 0, 0)):
        lo = (C.c_double * 3)(0.0, 0.0, 0.0)
        hi = (C.c_double * 3)(*box)
        u = (C.c_int * 3)(*user)
        out = (C.c_int * 3)()
        rc = L.sf_procgrid_choose(world, lo, hi, u, out)
        return tuple(out) if rc == 0 else None

    assert grid(8, (1.0, 1.0, 1.0)) == (2, 2, 2)
    assert grid(8, (8.0, 1.0, 1.0)) == (8, 1, 1)
    assert grid(4, (1.0, 1.0, 1.0)) == (1, 2, 2)              # three equal surfaces: the first visited (px slowest)
    assert grid(4, (2.0, 2.0, 1.0)) == (2, 2, 1)
    assert grid(6, (3.0, 2.0, 1.0)) == (3, 2, 1)
    assert grid(8, (1.0, 1.0, 1.0), (0, 1, 0)) == (2, 1, 4)   # `* 1 *`: 2 1 4 before the equal 4 1 2
    assert grid(8, (4.0, 1.0, 2.0), (0, 1, 0)) == (4, 1, 2)
    assert grid(2, (1.0, 3.0, 1.0)) == (1, 2, 1)
    assert grid(7, (1.0, 1.0, 1.0), (2, 0, 0)) is None        # Bad grid of processors
    assert grid(1, (1.0, 1.0, 1.0)) == (1, 1, 1)


def test_bench_watchdog_ends_a_job_that_does_not_finish():
    """bench.py --watchdog: a rank still running after the given time ends the job with rc 124 (an N > 1 run whose collective
    never completes must not hold the node); started before anything that could block, so it fires here without a GPU"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--watchdog", "0.2"], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 124 and "--watchdog" in r.stderr
    # ... and rank 0 still prints ONE parseable line that says what hung and where
    import json
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] is None and "watchdog" in line["error"] and line["stage"] and line["n_gpus"] == 1


def test_bench_prints_an_error_line_when_a_multi_gpu_run_cannot_come_up():
    """bench.py --gpus 2 as one rank of a world of 2 whose rendezvous never completes (nobody listens, short timeout): the run
    fails BEFORE it has a number and still prints one JSON line with `error` and the failing `stage` (a first hardware
    N > 1 record that does not come up must be diagnosable from the driver's record alone)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] is None and line["n_gpus"] == 2 and line["steps"] == 3 and line["error"] and line["stage"]


def test_bench_launches_its_own_ranks_and_still_prints_one_line():
    """plain `python3 bench.py --gpus 2` (the shape of the driver's command, no torch.distributed.run): bench.py starts the two
    ranks itself; here, without a GPU, both refuse -- the caller still gets exactly ONE parseable line (n_gpus 2, error +
    stage) and a non-zero exit code, never a bare launcher exit"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["value"] is None and line["n_gpus"] == 2 and line["steps"] == 3 and line["error"] and line["stage"]
