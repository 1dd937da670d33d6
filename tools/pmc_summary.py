#!/usr/bin/env python3
"""Reduce the rocprofv3 PMC passes of tools/pmc_hessian.sh to one tracked JSON (per launch, first launch skipped):
    python3 tools/pmc_summary.py gpurun_out/pmc_hess11008 11008 16 profiles/r02_hessian_C11008_pmc.json
HBM bytes as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE from SEPARATE passes, in KiB-like units of
1 KB; on gfx950 FETCH_SIZE counts the 128-B requests of wide coalesced reads at 64 B, so fetched bytes are doubled."""
import collections
import csv
import json
import os
import sys

csv.field_size_limit(1 << 30)
root, C, defer, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
KERNELS = {"kernel": "hessian16_big16_kernel", "fixup": "hessian16_big16_fixup"}


def mean_skip_first(v, per_launch=1):
    """Per-LAUNCH mean: a launch is `per_launch` consecutive dispatches (one per round of tiles); the first launch is
    the warm-up and is skipped."""
    v = v[per_launch:]
    return sum(v) / max(1, len(v) // per_launch)


counters, durations = collections.defaultdict(dict), {}
for grp in sorted(os.listdir(root)):
    path = os.path.join(root, grp, "pmc_counter_collection.csv")
    if not os.path.exists(path):
        continue
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    seen = set()
    for r in csv.DictReader(open(path)):
        for key, sub in KERNELS.items():
            if sub in r["Kernel_Name"]:
                vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if (key, r["Dispatch_Id"]) not in seen:
                    seen.add((key, r["Dispatch_Id"]))
                    dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    n_launch = max(1, len(dur.get("fixup", [])) or len(dur["kernel"]))
    for key in vals:
        per = max(1, len(dur[key]) // n_launch)                  # dispatches of this kernel per launch
        for c, v in vals[key].items():
            counters[key][c] = mean_skip_first(v, per)
        durations.setdefault(key, {})[grp] = mean_skip_first(dur[key], per)
        durations[key]["dispatches_per_launch"] = per

S = 2048
k, f = counters["kernel"], counters.get("fixup", {})
fetch = 2 * 1024 * (k["FETCH_SIZE"] + f.get("FETCH_SIZE", 0.0))
write = 1024 * (k["WRITE_SIZE"] + f.get("WRITE_SIZE", 0.0))
alg_bytes = defer * S * C * 2 + float(C) * C * 4          # X once + the upper half of H read and written
flops = defer * S * float(C) * C
cycles = k["GRBM_GUI_ACTIVE"] / 8.0                          # rocprofv3 sums the 8 XCDs
t_us = durations["kernel"]["SQ_VALU_MFMA_BUSY_CYCLES_SQ_BUSY_CYCLES_GRBM_GUI_ACTIVE"]
res = {
    "kernel": "hessian16_big16_kernel<f16> + hessian16_big16_fixup", "C": C, "samples_per_launch": defer, "tokens_per_sample": S,
    "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": alg_bytes,
    "FETCH_SIZE_KB_per_launch": k["FETCH_SIZE"] + f.get("FETCH_SIZE", 0.0),
    "WRITE_SIZE_KB_per_launch": k["WRITE_SIZE"] + f.get("WRITE_SIZE", 0.0),
    "fetch_bytes_corrected_x2": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
    "traffic_over_algorithmic": (fetch + write) / alg_bytes,
    "L2_hit_rate": k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"]),
    "kernel_us_per_launch_in_each_pass": durations["kernel"], "fixup_us_in_each_pass": durations.get("fixup", {}),
    "GRBM_GUI_ACTIVE_per_XCD": cycles, "effective_clock_GHz_profiled": cycles / t_us / 1e3,
    "SQ_VALU_MFMA_BUSY_CYCLES": k["SQ_VALU_MFMA_BUSY_CYCLES"],
    "mfma_busy_fraction_of_active_cycles": k["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024),
    "SQ_WAVE_CYCLES": k["SQ_WAVE_CYCLES"], "wait_any_frac": k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"],
    "wait_inst_any_frac": k["SQ_WAIT_INST_ANY"] / k["SQ_WAVE_CYCLES"], "active_inst_frac": k["SQ_ACTIVE_INST_ANY"] / k["SQ_WAVE_CYCLES"],
    "tflops_in_mfma_pass": flops / t_us / 1e6,
    "clock_adjusted_peak_tflops": 2500.0 * (cycles / t_us / 1e3) / 2.4,
}
res["fraction_of_clock_adjusted_peak"] = res["tflops_in_mfma_pass"] / res["clock_adjusted_peak_tflops"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
