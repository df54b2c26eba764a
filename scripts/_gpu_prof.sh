cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02/prof
for o in 1 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/prof/f$o -o f$o -- python scripts/fit_only.py 32768 3 --panel_fused=$o 2>&1 | grep "fit "
  f=$(find gpurun_out/r02/prof/f$o -name "*kernel_stats.csv" | head -1)
  echo "== panel_fused=$o: $f"; head -14 "$f" | cut -c1-160
done
