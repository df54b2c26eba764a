cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
W="python scripts/fit_only.py 32768 2 --gemm_tile=3"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/t3_fetch -o f -- $W > /dev/null 2>&1
python - <<'PY'
import csv, collections
agg=collections.defaultdict(float); cnt=collections.Counter()
for row in csv.DictReader(open('gpurun_out/t3_fetch/f_counter_collection.csv')):
    k=row['Kernel_Name'].split('(')[0]; agg[k]+=float(row['Counter_Value']); cnt[k]+=1
for k,v in agg.items():
    if v*2048/1e9>1: print(k, cnt[k], "launches, fetch GB per launch", v*1024*2/1e9/cnt[k])
PY
