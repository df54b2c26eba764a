# Round-5 profiles (run on the GPU box through gpurun, LAST, after the last change under friedrich_amd/csrc): bench line,
# rocprofv3 kernel stats of the bench command and of BASELINE configs[1], [2], [4], PMC passes (FETCH_SIZE / WRITE_SIZE / SQ
# counters, one pass per counter group, never together with a trace domain other than kernel-trace) of a fit at N = 32768 AND of
# configs[2] (N = 16384, Matern-5/2 + cholesky_epsilon: the configuration BASELINE.json names rocprof counters for), the
# baseline configs, gradient terms, configs[0], the sharded model terms.  Everything lands in gpurun_out/r05p/ and is copied
# into profiles/r05/ by hand afterwards (README.md there says what is what).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05p
mkdir -p $O
python scripts/source_hash.py > $O/sources.json
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 600 $O/bench_n1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -o bench -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_prof.json 2>/dev/null
cp $(find $O/bench_stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
for c in 1 2 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/config${c}_stats -o c$c -- python scripts/config_run.py $c > $O/config${c}_run.txt 2>/dev/null
  cp $(find $O/config${c}_stats -name "*kernel_stats.csv" | head -1) $O/config${c}_kernel_stats.csv
  cat $O/config${c}_run.txt | grep -v amdgpu
done
python scripts/baseline_configs.py 2>/dev/null | grep fit_ms > $O/baseline_final.jsonl
python scripts/grad_time.py 4096,8192,16384,32768 2>/dev/null | grep refactor > $O/grad_time.txt; cat $O/grad_time.txt
python scripts/config0_time.py 2>/dev/null | tail -1 > $O/config0_time.txt; cat $O/config0_time.txt
python scripts/dist_model.py 2>/dev/null > $O/dist_model.txt
# PMC passes: two fits at N = 32768 (bench.py's roofline.traffic), then configs[2]
for W in "fit32k|python scripts/fit_only.py 32768 2" "config2|python scripts/config_run.py 2"; do
  tag=${W%%|*}; cmd=${W#*|}
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o s -- $cmd > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -o f -- $cmd > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -o w -- $cmd > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/${tag}_sq -o q -- $cmd > /dev/null 2>&1
done
cp $(find gpurun_out/fit32k_stats -name "*kernel_stats.csv" | head -1) $O/fit32k_kernel_stats.csv
python scripts/summarise_counters.py fit32k $O/fit32k_counters.json "two fits (Gram + blocked Cholesky) at N=32768 d=16 RBF, automatic panel widths (2048 / 1024 / 512), scripts/fit_only.py 32768 2, one rocprofv3 --pmc pass per counter group" | head -5
python scripts/summarise_counters.py config2 $O/config2_counters.json "BASELINE configs[2]: N=16384 d=16 Matern-5/2 + cholesky_epsilon, fit x3 + predict(m=1024) x2 + predict_variance x2 (scripts/config_run.py 2), one rocprofv3 --pmc pass per counter group" | head -8
rm -rf $O/*_stats gpurun_out/fit32k_* gpurun_out/config2_*
ls $O
