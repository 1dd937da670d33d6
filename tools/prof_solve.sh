# usage (GPU box): bash tools/prof_solve.sh RxC [extra solve_probe flags] -- per-kernel stats of one solve shape
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_solve
rocprofv3 --kernel-trace -d $ROOT/gpurun_out/prof_solve -o x -- python3 $ROOT/tools/solve_probe.py "$@" --reps 2 > $ROOT/gpurun_out/prof_solve.log 2>&1
DB=$(ls $ROOT/gpurun_out/prof_solve/*/x_results.db $ROOT/gpurun_out/prof_solve/x_results.db 2>/dev/null | head -1)
python3 $ROOT/tools/solve_trace.py $DB | head -${LINES_OUT:-30}
rm -rf $ROOT/gpurun_out/prof_solve
