cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for o in 1 0; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/y$o -o y -- python scripts/fit_only.py 16384 3 --k4_yield=$o 2>&1 | grep "fit " | tail -1
python - <<PY
import csv
for i,row in enumerate(csv.DictReader(open('gpurun_out/y$o/y_kernel_stats.csv'))):
    if i<6: print(f"  {row['Name'][:50]:50s} calls {row['Calls']:>5s} total {float(row['TotalDurationNs'])/1e6:9.2f} ms avg {float(row['AverageNs'])/1e3:9.1f} us")
PY
done
