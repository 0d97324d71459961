"""Rewrite the round-6 table of DESIGN.md section 5 (between the r06-table markers) from profiles/r06_bench.json (the driver's
command on the round's final library), profiles/r06_bench_box2.json (the same command on another box of the pool) and
profiles/r06_pmc_summary.json.  usage: python tests/profile_readme_r06.py"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench.json")).readline())
x = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_box2.json")).readline())
d = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_summary.json")))
r, rx, f, fx, c, cx, cfg = b["roofline"], x["roofline"], b["fluidised_bed"], x["fluidised_bed"], b["configs"], x["configs"], b["config"]
h, cb, p = d["hbm_bytes_per_launch"], b["cpu_baseline"], b["parity"]
rows = [
    ("`value` (whole run, %d × 50 sub-steps, %d list rebuilds inside)" % (b["steps"], cfg["neighbor_rebuilds_in_run"]),
     "**%.2fe9 particle-substeps/s**, %.2f ms per 50 sub-steps (another box of the pool, `r06_bench_box2.json`: %.2fe9, %.2f ms)"
     % (b["value"] / 1e9, b["ms_per_step"], x["value"] / 1e9, x["ms_per_step"])),
    ("`k_substep<hertz>` mean launch (HIP events on the engine's stream, %d launches) / rocprofv3 kernel trace (`r06_kernel_stats.csv`)" % r["launches_timed"],
     "**%.1f µs** (box 2: %.1f; the boxes of the round: 174.4–189.9) / %.1f µs over the %d executed launches under the tracer"
     % (r["mean_kernel_us"], rx["mean_kernel_us"], d["kernel_trace"]["avg_us_executed"], d["kernel_trace"]["executed"])),
    ("`roofline` (594.35 B × 1 000 188 ÷ launch ÷ 8 TB/s)",
     "%.2f TB/s, **frac %.3f** (box 2: %.3f; round: 0.391–0.426); `frac_2m` %.3f (box 2: %.3f)"
     % (r["achieved"] / 1e3, r["frac"], rx["frac"], r["frac_2m"], rx["frac_2m"])),
    ("HBM traffic per launch (PMC, calibrated: `r06_pmc_summary.json`, `r06_traffic_calibration.json`; hash-matched to the library)",
     "%.1f MB = %.3f × algorithmic (read %.1f MB after the × 2 `FETCH_SIZE` correction, written %.1f MB); `TCC_MISS` × 128 B = %.1f MB"
     % (h["total_calibrated"] / 1e6, h["total_calibrated"] / h["algorithmic"], h["read_calibrated"] / 1e6, h["write"] / 1e6, h["tcc_miss_x_128B"] / 1e6)),
    ("list rebuild, packed bed (two-lane list build + sort + permutation)", "%.2f ms (round 5: 0.78)" % cfg["neighbor_rebuild_ms"]),
    ("loose bed (`fluidised_bed`: spacing 1.1 d, jitter 0.3 d; %d rebuilds in %d sub-steps)" % (f["neighbor_rebuilds_in_run"], 50 * f["steps"]),
     "**%.2fe9 /s**, kernel %.1f µs (frac %.3f on its own bytes), **whole run %.3f**, rebuild %.2f ms (box 2 over round 5's five steps: %.2fe9 /s, whole run **%.3f**, "
     "rebuild %.2f ms; round 5: 3.98e9, 0.266 / 0.246 on a slow box, 0.74 ms)"
     % (f["value"] / 1e9, f["mean_kernel_us"], f["roofline_frac"], f["roofline_frac_whole_run"], f["neighbor_rebuild_ms"], fx["value"] / 1e9,
        fx["roofline_frac_whole_run"], fx["neighbor_rebuild_ms"])),
    ("coupled step (ErgunWenYu + 50 sub-steps + scatter + Asrc + diffusion smoothing, 29×32×29 mesh)", "**%.1f steps/s**" % cfg["coupled_steps_per_s"]),
    ("host boundary (put_local_info + step(50) + get_local_info on host arrays, PCIe inclusive)", "%.2fe9 /s" % (cfg["host_boundary_substeps_per_s"] / 1e9)),
    ("`configs.C2` (10 080 grains, 32³ mesh, coupled)", "%.2fe8 /s, kernel %.1f µs (frac %.3f: launch-bound), %d coupled steps/s"
     % (c["C2"]["value"] / 1e8, c["C2"]["mean_kernel_us"], c["C2"]["roofline_frac"], round(c["C2"]["coupled_steps_per_s"]))),
    ("`configs.C3` (100 440-grain fluidised bed, 12×18×14 mesh, coupled; %d rebuilds in %d sub-steps)" % (c["C3"]["neighbor_rebuilds_in_run"], 50 * c["C3"]["steps"]),
     "**%.2fe9 /s**, kernel %.1f µs (frac %.3f), **whole run %.3f** (box 2 over round 5's five steps: %.3f; round 5: 2.01e9, 0.133), %d coupled steps/s"
     % (c["C3"]["value"] / 1e9, c["C3"]["mean_kernel_us"], c["C3"]["roofline_frac"], c["C3"]["roofline_frac_whole_run"],
        cx["C3"]["roofline_frac_whole_run"], round(c["C3"]["coupled_steps_per_s"]))),
    ("`configs.C5` (500 k polydisperse d = 0.85–1.0 mm, hertzFix/history + lubricate/poly + fix cohesive)",
     "%.2fe9 /s, kernel %.1f µs (frac %.3f), whole run %.3f" % (c["C5"]["value"] / 1e9, c["C5"]["mean_kernel_us"], c["C5"]["roofline_frac"], c["C5"]["roofline_frac_whole_run"])),
    ("`configs.C5_wide` (500 k grains d ~ U(0.5, 1.5) mm, SURVEY §8(d)'s arguments; 10 sub-steps per step, `profiles/r06_README.md` §4)",
     "%.2fe9 /s, kernel %.1f µs (frac %.3f; K_half %.2f, longest row %d)"
     % (c["C5_wide"]["value"] / 1e9, c["C5_wide"]["mean_kernel_us"], c["C5_wide"]["roofline_frac"], c["C5_wide"]["k_half"], c["C5_wide"]["longest_row"])),
    ("CPU baseline, oracle single thread, same bed", "%.2fe6 /s; coupled step %.3f steps/s; 16 independent processes: %.1fe7 /s"
     % (cb["value"] / 1e6, cb["coupled_steps_per_s"], b["cpu_baseline_all_cores"]["value"] / 1e7)),
    ("GPU vs oracle on the whole bed after 50 sub-steps (`parity`)", "tags identical, max abs(Δx) %.1e d, v %.1e, ω %.1e, f %.1e relative"
     % (p["max_abs_dx_over_d"], p["max_rel_v"], p["max_rel_omega"], p["max_rel_f"])),
]
tab = "| quantity | value |\n|---|---|\n" + "\n".join("| %s | %s |" % rw for rw in rows)
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
b0, b1 = "<!-- r06-table-begin -->", "<!-- r06-table-end -->"
i0, i1 = s.index(b0) + len(b0), s.index(b1)
open(path, "w").write(s[:i0] + "\n" + tab + "\n" + s[i1:])
print(tab)
