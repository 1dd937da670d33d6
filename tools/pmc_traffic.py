#!/usr/bin/env python3
"""HBM traffic per Hessian launch from two rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM / rocprofv3 section:
FETCH_SIZE and WRITE_SIZE are in KiB-like units of 1 KB, collected in SEPARATE --pmc passes; on gfx950
FETCH_SIZE counts 64-B requests as 32 B, so the fetched bytes are doubled).

    python3 tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> \
            [--kernels hessian16_big16_kernel,hessian16_big16_fixup] [--launch-of hessian16_big16_kernel] [--shape "..."]
Counters of all `--kernels` dispatches are summed and divided by the number of `--launch-of` dispatches
(one Hessian update = one big kernel + its fixup), skipping the first (warm-up) launch."""
import argparse
import csv
import json
import sys

csv.field_size_limit(sys.maxsize)


def collect(path, counter, kernels, launch_of):
    total, launches, first_skipped = 0.0, 0, set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        hit = [k for k in kernels if k in name]
        if not hit:
            continue
        if hit[0] not in first_skipped:          # warm-up launch of each kernel
            first_skipped.add(hit[0])
            continue
        total += float(r["Counter_Value"])
        if launch_of in name:
            launches += 1
    return total, launches


ap = argparse.ArgumentParser()
ap.add_argument("fetch_csv")
ap.add_argument("write_csv")
ap.add_argument("out")
ap.add_argument("--kernels", default="hessian16_big16_kernel,hessian16_big16_fixup")
ap.add_argument("--launch-of", default="hessian16_big16_kernel")
ap.add_argument("--samples", type=int, default=16, help="2048-token samples folded per launch (C = 8192)")
a = ap.parse_args()
ks = a.kernels.split(",")
f, nf = collect(a.fetch_csv, "FETCH_SIZE", ks, a.launch_of)
w, nw = collect(a.write_csv, "WRITE_SIZE", ks, a.launch_of)
a.shape = f"C=8192, S=2048 tokens x {a.samples} samples per launch, fp16"
a.algorithmic_bytes = a.samples * 2048 * 8192 * 2 + 2 * 8192 * 8192 * 4 / 2
out = {
    "kernels": ks, "shape": a.shape, "samples_per_launch": a.samples,
    "FETCH_SIZE_KB_per_launch": f / max(nf, 1), "FETCH_SIZE_launches": nf,
    "WRITE_SIZE_KB_per_launch": w / max(nw, 1), "WRITE_SIZE_launches": nw,
    "fetch_bytes_corrected_x2": 2 * 1024 * f / max(nf, 1),
    "write_bytes": 1024 * w / max(nw, 1),
    "algorithmic_bytes_per_launch": a.algorithmic_bytes,
}
out["hbm_bytes_per_launch"] = out["fetch_bytes_corrected_x2"] + out["write_bytes"]
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(out, indent=1))
