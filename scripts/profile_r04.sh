# Round-4 profiles (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats of the bench command and of
# BASELINE configs[1], [2], [4], PMC passes (FETCH_SIZE / WRITE_SIZE / SQ counters, one pass per counter group) of a fit, and
# FETCH_SIZE of configs[4] before / after the big solve leaves (bigleaf_max = 0 restores round 3's column groups).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04p
mkdir -p $O
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 600 $O/bench_n1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -o bench -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_prof.json 2>/dev/null
cp $(find $O/bench_stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
for c in 1 2 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/config${c}_stats -o c$c -- python scripts/config_run.py $c > $O/config${c}_run.txt 2>/dev/null
  cp $(find $O/config${c}_stats -name "*kernel_stats.csv" | head -1) $O/config${c}_kernel_stats.csv
  cat $O/config${c}_run.txt | grep -v amdgpu
done
# configs[4] with round 3's solve path (column groups of 16) for the before / after comparison: kernel stats + FETCH_SIZE of both
rocprofv3 --kernel-trace --stats --output-format csv -d $O/config4b_stats -o c4b -- python scripts/config_run.py 4 --bigleaf_max=0 > $O/config4_round3path_run.txt 2>/dev/null
cp $(find $O/config4b_stats -name "*kernel_stats.csv" | head -1) $O/config4_round3path_kernel_stats.csv
grep -v amdgpu $O/config4_round3path_run.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/c4after_fetch -o f -- python scripts/config_run.py 4 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/c4before_fetch -o f -- python scripts/config_run.py 4 --bigleaf_max=0 > /dev/null 2>&1
python - <<'PY' > gpurun_out/r04p/config4_fetch_before_after.json
import collections, csv, glob, json
out = {"what": "configs[4] (4096 rows + 8 x add_samples(512) + sample_at(256), twice): rocprofv3 --pmc FETCH_SIZE per kernel, x2 (gfx950 correction), GB summed over all launches; 'before' = bigleaf_max=0 (round 3's column groups of 16), 'after' = the default path", "GB": {}}
for tag in ("before", "after"):
    fs = glob.glob(f"gpurun_out/c4{tag}_fetch/**/*_counter_collection.csv", recursive=True)
    agg = collections.defaultdict(float)
    for row in csv.DictReader(open(fs[0])):
        if row["Counter_Name"] == "FETCH_SIZE":
            agg[row["Kernel_Name"].split("(")[0].replace("void ", "")] += float(row["Counter_Value"]) * 1024 * 2 / 1e9
    out["GB"][tag] = {k: round(v, 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]) if v > 0.01}
    out["GB"][tag + "_total"] = round(sum(agg.values()), 3)
print(json.dumps(out, indent=1))
PY
cat gpurun_out/r04p/config4_fetch_before_after.json | head -40
python scripts/baseline_configs.py 2>/dev/null | grep fit_ms > $O/baseline_final.jsonl
python scripts/grad_time.py 4096,8192,16384,32768 2>/dev/null | grep refactor > $O/grad_time.txt; cat $O/grad_time.txt
python scripts/config0_time.py 2>/dev/null | tail -1 > $O/config0_time.txt; cat $O/config0_time.txt
python scripts/narrow_wide_ab.py 4096,8192 2>/dev/null | grep -v amdgpu > $O/solve_paths_ab.txt
# PMC passes on two fits at N = 32768 (separate passes, never together with a trace domain other than kernel-trace)
W="python scripts/fit_only.py 32768 2"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fit32k_stats -o s -- $W > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fit32k_fetch -o f -- $W > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/fit32k_write -o w -- $W > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/fit32k_sq -o q -- $W > /dev/null 2>&1
cp $(find gpurun_out/fit32k_stats -name "*kernel_stats.csv" | head -1) $O/fit32k_kernel_stats.csv
python scripts/summarise_counters.py fit32k $O/fit32k_counters.json "two fits (Gram + blocked Cholesky) at N=32768 d=16 RBF nb=1024, scripts/fit_only.py 32768 2, one rocprofv3 --pmc pass per counter group" | head -5
python scripts/dist_model.py 2>/dev/null > $O/dist_model.txt
rm -rf $O/*_stats gpurun_out/fit32k_* gpurun_out/c4*_fetch
ls $O
