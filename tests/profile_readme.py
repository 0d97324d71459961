"""Write profiles/<tag>_README.md + <tag>_kernel_stats.csv from the outputs of one GPU call (see profiles/*_README.md):
gpurun_out/kt_<tag>/p_kernel_stats.csv, profiles/<tag>_pmc_summary.json (tests/pmc_summarize.py), gpurun_out/bench_<tag>.json
usage: python tests/profile_readme.py TAG "title line" """
import csv
import json
import shutil
import sys

tag, title = sys.argv[1], sys.argv[2]
shutil.copy(f"gpurun_out/kt_{tag}/p_kernel_stats.csv", f"profiles/{tag}_kernel_stats.csv")
d = json.loads(open(f"gpurun_out/bench_{tag}.json").read().strip().splitlines()[-1])
s = json.load(open(f"profiles/{tag}_pmc_summary.json"))
d["roofline"]["traffic"] = s["hbm_bytes_per_launch"].get("total_calibrated", s["hbm_bytes_per_launch"]["total_raw"])
d["roofline"]["traffic_source"] = f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc, bytes per launch)"
kt, v = s["kernel_trace"], s["vector_l1_model"]
rows = list(csv.DictReader(open(f"profiles/{tag}_kernel_stats.csv")))


def avg(name):
    for r in rows:
        if name in r["Name"]:
            return float(r["AverageNs"]) / 1e3
    return 0.0


sort_us = avg("k_cell_count") + avg("k_cell_place") + avg("k_cell_sort_segments")
text = f"""# {tag} -- {title}

`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline`
-> `{tag}_kernel_stats.csv`.  `k_substep<2,false,false,1>`: {kt['launches']} launches, rocprofv3 average {kt['avg_us_all']:.1f} us over ALL
launches; {kt['launches'] - kt['executed']} of them are 3-5 us early exits, the {kt['executed']} executed launches average **{kt['avg_us_executed']:.1f} us**; bench.py's
HIP-event mean over the sampled executed launches (every 8th sub-step) in the un-profiled default run on the same box:
**{d['roofline']['mean_kernel_us']:.1f} us** (line below; the boxes of this pool differ by up to 8 % on this kernel, 197-217 us).
Other kernels of one neighbour rebuild: k_build_neigh {avg('k_build_neigh'):.0f} us (394 in r01_g), counting sort
k_key_count + k_key_place + k_key_rank {avg('k_key_count') + avg('k_key_place') + avg('k_key_rank'):.0f} us; of one coupled step: k_drag_on_particles {avg('k_drag_on_particles'):.0f} us,
k_calc_tc<8> {avg('k_calc_tc'):.0f} us (48), k_particle_to_eulerian<8> {avg('k_particle_to_eulerian'):.0f} us (26), particles by cell (k_cell_count + k_cell_place +
k_cell_sort_segments) {sort_us:.0f} us (62 in r01_g), k_spectral_pass {avg('k_spectral_pass'):.1f} us x 18.

PMC passes (`tests/pmc.sh {tag} ...`, one `--pmc` set per run, `--kernel-trace` only, every pass under `timeout`; folded by
`tests/pmc_summarize.py`) -> `{tag}_pmc_summary.json`: raw counters FETCH_SIZE {s['hbm_bytes_per_launch']['read_raw'] / 1e6:.0f} MB + WRITE_SIZE {s['hbm_bytes_per_launch']['write'] / 1e6:.0f} MB per
launch; calibrated on known byte counts in the same call (`tests/calibrate_traffic.sh`: FETCH_SIZE = 1/2 of the bytes read for every
load width, WRITE_SIZE = 1, TCC_MISS_sum = 128-byte lines read + written) the kernel reads {s['hbm_bytes_per_launch'].get('read_calibrated', 0) / 1e6:.0f} MB and writes {s['hbm_bytes_per_launch']['write'] / 1e6:.0f} MB =
**{s['hbm_bytes_per_launch'].get('total_calibrated', 0) / 1e6:.0f} MB per launch** (TCC_MISS_sum x 128 B = {s['hbm_bytes_per_launch'].get('tcc_miss_x_128B', 0) / 1e6:.0f} MB; algorithmic 594 MB); {v['l2_read_requests_per_particle']:.1f} L2 read requests per particle, mean latency {v['mean_l2_read_latency_cycles']:.0f} cycles, {v['mean_outstanding_read_requests_per_cu']:.0f} read requests in
flight per CU; L2 hit rate {s['l2_hit_rate']:.2f}; wave cycles: {100 * s['wave_cycle_split']['waiting_any']:.0f} % waiting on memory, {100 * s['wave_cycle_split']['issue_stall']:.0f} % issue stall, {100 * s['wave_cycle_split']['issuing']:.0f} % issuing.
The store / load experiments that price the history traffic are in profiles/r01_f_README.md.

bench.py (default command, same code, same box):
{json.dumps(d)}
"""
open(f"profiles/{tag}_README.md", "w").write(text)
print(text[:1500])
