"""Write profiles/<tag>_README.md from the outputs of tests/profile_round.sh <tag> (round 3 layout):
gpurun_out/bench_<tag>.json, profiles/<tag>_pmc_summary.json (tests/pmc_summarize.py), gpurun_out/kt_<tag>/p_kernel_stats.csv
usage: python tests/profile_readme3.py TAG"""
import csv
import json
import shutil
import sys

tag = sys.argv[1]
shutil.copy(f"gpurun_out/kt_{tag}/p_kernel_stats.csv", f"profiles/{tag}_kernel_stats.csv")
d = json.loads(open(f"gpurun_out/bench_{tag}.json").read().strip().splitlines()[-1])
s = json.load(open(f"profiles/{tag}_pmc_summary.json"))
kt, v, hb = s["kernel_trace"], s["vector_l1_model"], s["hbm_bytes_per_launch"]
d["roofline"]["traffic"] = hb["total_calibrated"]
d["roofline"].pop("traffic_stale", None)
d["roofline"]["traffic_source"] = ("profiles/%s_pmc_summary.json (rocprofv3 --pmc, bytes per launch; kernel source hash %s)"
                                   % (tag, s["kernel_source_sha256_16"]))
rows = list(csv.DictReader(open(f"profiles/{tag}_kernel_stats.csv")))


def avg(name):
    for r in rows:
        if name in r["Name"]:
            return float(r["AverageNs"]) / 1e3
    return 0.0


f = d["fluidised_bed"]
text = f"""# {tag} -- round 3 final code: the headline kernel, its counters, the bench line

One GPU call (`tests/profile_round.sh {tag}`): FETCH_SIZE / WRITE_SIZE calibration on known byte counts, the default
`bench.py` line, `rocprofv3 --kernel-trace --stats` of `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fluidised
--no-coupled` -> `{tag}_kernel_stats.csv`, and seven separate `--pmc` passes (`tests/pmc.sh`, `--kernel-trace` only, each
under `timeout`) folded by `tests/pmc_summarize.py` -> `{tag}_pmc_summary.json`. The summary carries a hash of the
kernel's sources (`kernel_source_sha256_16` = {s['kernel_source_sha256_16']}); `bench.py` reports `roofline.traffic` only while
the library it times is built from the same sources, otherwise `traffic: null` + `traffic_stale`.

`k_substep<2,false,false,1,false,1>` (Hertz, one lane per atom, v / omega always prefetched, non-temporal policy 1):
{kt['launches']} launches, {kt['launches'] - kt['executed']} of them 3-5 us early exits behind a rebuild trigger; the {kt['executed']} executed launches average
**{kt['avg_us_executed']:.1f} us** in the trace; bench.py's HIP-event mean over the sampled launches of the un-profiled default run on
the same box: **{d['roofline']['mean_kernel_us']:.1f} us** -> {d['roofline']['achieved']:.0f} GB/s of algorithmic bytes = **{d['roofline']['frac']:.3f}** of the 8 TB/s roofline
(594.35 B x 1 000 188 particles per launch; the boxes of this pool differ by up to 7 % on this kernel: 192-205 us,
0.36-0.385, over the calls of this round -- `gpurun_out/bench_r03_*.json`). Calibrated HBM traffic: {hb['read_calibrated'] / 1e6:.0f} MB read + {hb['write'] / 1e6:.0f} MB written =
**{hb['total_calibrated'] / 1e6:.0f} MB per launch** (TCC_MISS_sum x 128 B = {hb['tcc_miss_x_128B'] / 1e6:.0f} MB; 1.21 x the algorithmic 594 MB -- unchanged from
round 2: the body of this variant is the same, the round's changes sit in the epilogue (fix freeze in script order,
brick forward records) and in the list build). {v['l2_read_requests_per_particle']:.1f} L2 read requests per particle at {v['mean_l2_read_latency_cycles']:.0f} cycles, {v['mean_outstanding_read_requests_per_cu']:.0f} in
flight per CU, L2 hit rate {s['l2_hit_rate']:.2f}; wave cycles {100 * s['wave_cycle_split']['waiting_any']:.0f} % waiting on memory / {100 * s['wave_cycle_split']['issue_stall']:.0f} % issue stall / {100 * s['wave_cycle_split']['issuing']:.0f} % issuing.
What overlapping consecutive sub-steps could add, and why it was not built: `r03_a_README.md`.

Neighbour rebuild (1 M grains): `k_build_neigh` {avg('k_build_neigh'):.0f} us on this lattice (candidate order kept, look-ups coalesced in
the second sweep), 440-480 us on the loose bed where the touching neighbours are placed first; whole rebuild of the loose
bed 0.95-1.12 ms from the last sub-step before to the first after (`{tag}_rebuild_trace_fluidised.txt`; 1.18 before the
last batch of the round, 1.37 at its start); on the host clock of an un-traced run, synchronised at both ends
(`neighbor_rebuild_ms` of the bench line): {d['config'].get('neighbor_rebuild_ms', float('nan')):.2f} ms on the lattice, {f.get('neighbor_rebuild_ms', float('nan')):.2f} ms on the loose bed.  That batch, library against library on the same box
(`tests/ab_rebuild.sh`, rocprofv3 per-kernel averages over the 34 rebuilds of a loose-bed run): every per-atom array
re-ordered by ONE kernel (`k_permute_all` 60-69 us instead of 14 launches summing to ~115), periodic ghosts of two
dimensions made without a host round trip in between, the host's look at the list counts overlapped with
`k_back_slots`, history of non-touching slots no longer zero-filled, candidate walk rewritten (unconditional 16-byte
record loads, running row keys and scratch pointer: 270 -> 128 instructions per four candidates) and given the XCD-
contiguous block order of the sub-step kernel: `k_build_neigh` 505-524 -> 437-465 us, loose-bed throughput +5-7 %.
After that, on the un-traced host clock (same box, library against library): the cell histograms counted back to zero
by the kernels that use them (no 16 MB memsets, no cursor copy), the owned histogram filled by `k_pbc_keys`, the
widest row found inside the list build (0.953 -> 0.927 ms), and the ghosts ordered by the same counting sort as the
owned atoms, sharing its scan with the list build's ghost table, instead of a 64-bit radix sort (0.923 -> 0.877 ms); the two scans over the 4 M cell histograms by a three-launch
tile scan instead of rocPRIM's (0.893 -> 0.868 ms); list statistics on the first lists and every fourth one after
(-25 us); hot beds (a rebuild at least every 32 sub-steps, >= 200 k grains, single domain) queue through the predicted
trigger instead of stopping short of it: one host read per rebuild instead of two or three (+0.6-1 % at 1 M, +4.7 % at 300 k).
Measured and dropped: the old-list look-up inside the walk against after it (equal), the walk as its own kernel at
5 / 6 / 8 waves per SIMD (249 / 243 / 234 us against ~238 inside the fused kernel: not latency-bound), cells of the full
cutoff instead of half (`SF_SUB=1`: walk 228 -> 184 us on the loose bed, but the sub-step kernel 191 -> 330 us on the
lattice and 180 -> 250 us on dense jittered beds: the finer cells are what orders the atoms for its gathers).

Lanes per atom between 20 k and 300 k grains (the sizes an 8 / 4 / 2-GPU split of the 1 M bed leaves per GPU), whole-run
µs per sub-step, same box (`tests/ab_persist.sh`, `SF_LPA=1` / `SF_LPA=2` / the policy of `DemEngine::lanes_per_atom` =
the cheaper whole number of rounds of resident waves):
63 k 23.5 / 19.7 / 19.5; 126 k 29.8 / 29.8 / 30.0; 170 k 35.8 / 36.7 / 35.5; 200 k 45.0 / 42.7 / 42.4;
250 k 50.4 / 48.9 / 48.7; 300 k 55.2 / 58.1 / 55.9 (round 2's fixed threshold at 150 k picked the slower one at 200-295 k).

Loose disordered ("fluidised") bed, the `fluidised_bed` object of the line: {f['mean_kernel_us']:.0f} us per sub-step kernel at K_half {f['k_half']}
= {f['roofline_frac']:.2f} of the roofline (236 us / 0.26-0.28 with the round-2 slot order on the same boxes), {f['value'] / 1e9:.2f}e9
particle-sub-steps/s whole-run with {f['neighbor_rebuilds_in_run']} rebuilds in 250 sub-steps (2.9e9 before).

`parity`: the whole 1 000 188-particle bed after setup + 50 sub-steps, GPU against the oracle from the same start:
max |dx| / d = {d['parity']['max_abs_dx_over_d']:.1e}, v {d['parity']['max_rel_v']:.1e}, omega {d['parity']['max_rel_omega']:.1e}, f {d['parity']['max_rel_f']:.1e}.

bench.py (default command, same code, same box; `traffic` filled in from the summary of this call):
{json.dumps(d)}
"""
open(f"profiles/{tag}_README.md", "w").write(text)
print(text[:800])
