#!/bin/bash
# PMC passes (separate runs per counter group) for quant_super_kernel: tools/pmc_qs.sh [RxC]   (on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SHAPE=${1:-4096x2048}
OUT=$ROOT/gpurun_out/pmc_qs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp GPTQ_LOOKAHEAD=0 GPTQ_QS_LANES=${GPTQ_QS_LANES:-16}
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  name=$(echo $grp | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$name -o pmc -- python3 $ROOT/tools/solve_probe.py $SHAPE --reps 1 > $OUT/$name.log 2>&1 || echo "pass $name failed"
  python3 - "$OUT/$name" <<'PY'
import csv, sys, glob, collections
csv.field_size_limit(sys.maxsize)
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][-40:]
    if "quant_super" in k or "trailing" in k or "quant_block" in k:
        a = agg[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    print(f"{k:42s} {c:22s} per launch {v / n:14.1f}  ({n} launches)")
PY
done
