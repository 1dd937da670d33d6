#!/bin/bash
# PMC passes (separate runs per counter group; rocprofv3 serialises kernels under --pmc, so every kernel is seen ALONE)
# for the kernels of one solve: tools/pmc_solve.sh [RxC] [extra solve_probe flags]   (on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SHAPE=${1:-4096x11008}; shift
OUT=$ROOT/gpurun_out/pmc_solve
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$name -o pmc -- python3 $ROOT/tools/solve_probe.py $SHAPE --reps 1 "$@" > $OUT/$name.log 2>&1 || echo "pass $name failed"
  python3 - "$OUT/$name" <<'PY'
import csv, sys, glob, collections
csv.field_size_limit(sys.maxsize)
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file in", sys.argv[1]); sys.exit(0)
agg = collections.defaultdict(lambda: [0.0, 0])
dur = collections.defaultdict(lambda: [0.0, set()])
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gptq::", "")[:34]
    if not any(s in k for s in ("syrk128", "trailing128", "chol_panel", "quant_super", "syrk_kernel", "trailing_kernel")):
        continue
    a = agg[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    d = dur[k]
    if r["Dispatch_Id"] not in d[1]:
        d[1].add(r["Dispatch_Id"]); d[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (k, c), (v, n) in sorted(agg.items()):
    print(f"{k:36s} {c:26s} sum {v:16.0f}  launches {n:4d}  kernel time {dur[k][0] / 1e3:8.3f} ms")
PY
done
