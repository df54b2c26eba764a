# Round-3 profiles (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats of the bench command and of
# BASELINE configs[1], [2], [4], PMC passes (FETCH_SIZE / WRITE_SIZE / SQ counters, one pass per counter group) of a fit.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03p
mkdir -p $O
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 3000 $O/bench_n1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -o bench -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_prof.json 2>/dev/null
cp $(find $O/bench_stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
for c in 1 2 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/config${c}_stats -o c$c -- python scripts/config_run.py $c > $O/config${c}_run.txt 2>/dev/null
  cp $(find $O/config${c}_stats -name "*kernel_stats.csv" | head -1) $O/config${c}_kernel_stats.csv
  cat $O/config${c}_run.txt | grep -v amdgpu
done
python scripts/baseline_configs.py 2>/dev/null | grep fit_ms > $O/baseline_final.jsonl
python scripts/grad_time.py 4096,8192,16384,32768 2>/dev/null | grep refactor > $O/grad_time.txt; cat $O/grad_time.txt
python scripts/config0_time.py 2>/dev/null | tail -1 > $O/config0_time.txt; cat $O/config0_time.txt
# PMC passes on two fits at N = 32768 (separate passes, never together with a trace domain other than kernel-trace)
W="python scripts/fit_only.py 32768 2"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fit32k_stats -o s -- $W > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fit32k_fetch -o f -- $W > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/fit32k_write -o w -- $W > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/fit32k_sq -o q -- $W > /dev/null 2>&1
ls gpurun_out/fit32k_*/*/ | head -20
python scripts/summarise_counters.py fit32k $O/fit32k_counters.json "two fits (Gram + blocked Cholesky) at N=32768 d=16 RBF nb=1024, scripts/fit_only.py 32768 2, one rocprofv3 --pmc pass per counter group"
python scripts/dist_model.py 2>/dev/null > $O/dist_model.txt; cat $O/dist_model.txt
