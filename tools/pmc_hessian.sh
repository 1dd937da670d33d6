#!/bin/bash
# PMC passes for the Hessian kernel at one launch shape (separate rocprofv3 runs per counter group, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
#   tools/pmc_hessian.sh C DEFER OUTDIR      (run on the GPU box; writes OUTDIR/<group>/..._counter_collection.csv)
set -e
C=${1:-11008}; DEFER=${2:-16}; OUT=${3:-gpurun_out/pmc_hess}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOT/$OUT/$name" -o pmc -- \
      python3 "$ROOT/tools/hessian_probe.py" --defer $DEFER $C > "$ROOT/$OUT/$name.log" 2>&1 || echo "pass $name failed"
  grep "TFLOP" "$ROOT/$OUT/$name.log" || true
done
