# round-3 evidence: bench line, rocprofv3 kernel stats of the same command, Hessian dispatches, solve trace, PMC traffic
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $ROOT/gpurun_out/r03b_bench_full.json 2> $ROOT/gpurun_out/r03b_bench_full.err
tail -c 600 $ROOT/gpurun_out/r03b_bench_full.json
rm -rf $ROOT/gpurun_out/prof_b
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_b -o x -- python3 $ROOT/bench.py --steps 5 --no-cpu-baseline --no-also --end-to-end 0 > $ROOT/gpurun_out/r03b_bench_profiled_run.json 2> $ROOT/gpurun_out/r03b_bench_profiled_run.err
DB=$(ls $ROOT/gpurun_out/prof_b/*/x_results.db $ROOT/gpurun_out/prof_b/x_results.db 2>/dev/null | head -1)
python3 $ROOT/tools/kernel_stats_csv.py $DB $ROOT/gpurun_out/r03b_bench_kernel_stats.csv --dispatches hessian16_big16 $ROOT/gpurun_out/r03b_bench_hessian_dispatches.csv
python3 $ROOT/tools/solve_trace.py $DB > $ROOT/gpurun_out/r03b_solve_trace_down_proj.txt
head -12 $ROOT/gpurun_out/r03b_bench_kernel_stats.csv | cut -c1-140
head -30 $ROOT/gpurun_out/r03b_solve_trace_down_proj.txt
rm -rf $ROOT/gpurun_out/prof_b
bash $ROOT/tools/pmc_hessian.sh 11008 16 gpurun_out/pmc_hess_r03 > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/pmc_hess_r03 11008 16 $ROOT/gpurun_out/r03b_hessian_C11008_pmc.json 2>&1 | tail -3
ls $ROOT/gpurun_out/pmc_hess_r03 | head
