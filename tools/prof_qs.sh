# usage: bash tools/prof_qs.sh   (on the GPU box) -- kernel durations of the solve per GPTQ_QS_LANES mode, helper stream on / off
cd /tmp && export TMPDIR=/tmp
for cfg in "16 3" "16 0" "8 0" "0 0"; do
  set -- $cfg
  export GPTQ_QS_LANES=$1 GPTQ_LOOKAHEAD=$2
  tag=qs$1_la$2
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o x -- python3 $GRAFT_REPO_ROOT/tools/solve_probe.py ${SHAPE:-4096x11008} --actorder --reps 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/kernel_stats_csv.py $(ls $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/*/x_results.db $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/x_results.db 2>/dev/null | head -1) $GRAFT_REPO_ROOT/gpurun_out/r03_${tag}_kernel_stats.csv
  echo "== lanes $1 lookahead $2"; grep fasterquant $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log
  grep -E "quant_super|quant_block|trailing_kernel|potrf|syrk_kernel|panel" $GRAFT_REPO_ROOT/gpurun_out/r03_${tag}_kernel_stats.csv | cut -d, -f1-4,6,7 | sed 's/(gptq::QuantSuperArgs)//;s/(float.*)"/"/'
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$tag
done
