# usage (GPU box): bash tools/prof_idle.sh -- idle windows of the GPU and the phases of the last bench step (kernel trace of a 3-step run)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_idle
rocprofv3 --kernel-trace -d $ROOT/gpurun_out/prof_idle -o x -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-also --end-to-end 0 > $ROOT/gpurun_out/prof_idle.json 2> $ROOT/gpurun_out/prof_idle.err
DB=$(ls $ROOT/gpurun_out/prof_idle/*/x_results.db $ROOT/gpurun_out/prof_idle/x_results.db 2>/dev/null | head -1)
python3 $ROOT/tools/idle_gaps.py $DB --last-ms ${LAST_MS:-80} --min-us 20 | head -8
python3 $ROOT/tools/step_phases.py $DB
rm -rf $ROOT/gpurun_out/prof_idle
