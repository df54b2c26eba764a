# Round-6 profiles (run on the GPU box through gpurun, LAST, after the last change under friedrich_amd/csrc): bench line, rocprofv3
# kernel stats of the bench command and of BASELINE configs[1], [2], [4], PMC passes (FETCH_SIZE / WRITE_SIZE / SQ counters, one pass per
# counter group, never together with a trace domain other than kernel-trace) of a fit at N = 32768 and of configs[2]; the baseline
# configs, gradient terms (+ per-kernel stats at N = 4096 / 32768), configs[0], the sharded model terms; the resident panel chain:
# stamps, in-process A/B against the launch chain, timelines and step periods.  The fits under --pmc run with panel_chain = 0 where
# noted: a tool that serialises kernels is fine for the resident launch (its waits are inside ONE launch), the option only keeps the
# counter passes comparable with round 5's.  Everything lands in gpurun_out/r06p/ and is copied into profiles/r06/ afterwards.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06p
mkdir -p $O
python scripts/source_hash.py > $O/sources.json
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 400 $O/bench_n1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -o bench -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_prof.json 2>/dev/null
cp $(find $O/bench_stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
for c in 1 2 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/config${c}_stats -o c$c -- python scripts/config_run.py $c > $O/config${c}_run.txt 2>/dev/null
  cp $(find $O/config${c}_stats -name "*kernel_stats.csv" | head -1) $O/config${c}_kernel_stats.csv
  cat $O/config${c}_run.txt | grep -v amdgpu
done
python scripts/baseline_configs.py 2>/dev/null | grep fit_ms > $O/baseline_final.jsonl
python scripts/baseline_configs.py 4096,8192,16384 --panel_chain=0 2>/dev/null | grep fit_ms > $O/baseline_launch_chain.jsonl
python scripts/grad_time.py 4096,8192,16384,32768 2>/dev/null | grep refactor > $O/grad_time.txt; cat $O/grad_time.txt
python scripts/config0_time.py 2>/dev/null | tail -1 > $O/config0_time.txt; cat $O/config0_time.txt
python scripts/dist_model.py 2>/dev/null > $O/dist_model.txt
python scripts/panel_chain_probe.py 256,512,1024,2048,3072,4096,6144,8192,12288,16384 --mode=2 2>/dev/null | grep launch > $O/panel_chain_ab.txt; cat $O/panel_chain_ab.txt
python scripts/panel_chain_stamps.py 512 2>/dev/null | grep -v amdgpu > $O/panel_chain_stamps.txt
scripts/waitvalue_probe > $O/waitvalue_probe.txt 2>&1
for n in 4096 8192 16384; do
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python scripts/fit_only.py $n 3 > /dev/null 2>&1
  T=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
  python scripts/trace_timeline.py $T 0.40 0.58 > $O/fit${n}_timeline.txt
  python scripts/step_periods.py $T $n > $O/steps${n}.txt
  rm -rf gpurun_out/tl
done
# PMC passes: two fits at N = 32768 (bench.py's roofline.traffic), then configs[2]
for W in "fit32k|python scripts/fit_only.py 32768 2" "config2|python scripts/config_run.py 2"; do
  tag=${W%%|*}; cmd=${W#*|}
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o s -- $cmd > /dev/null 2>&1
  # (every counter pass under its own time limit: the pass of round 6's fused build -- history: tag fused-la-opt-in -- never
  # finished and took the whole call's 3000 s with it)
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -o f -- $cmd > /dev/null 2>&1 || echo "PMC pass FETCH_SIZE of $tag: rc $?"
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -o w -- $cmd > /dev/null 2>&1 || echo "PMC pass WRITE_SIZE of $tag: rc $?"
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/${tag}_sq -o q -- $cmd > /dev/null 2>&1 || echo "PMC pass SQ of $tag: rc $?"
done
cp $(find gpurun_out/fit32k_stats -name "*kernel_stats.csv" | head -1) $O/fit32k_kernel_stats.csv
python scripts/summarise_counters.py fit32k $O/fit32k_counters.json "two fits (Gram + blocked Cholesky) at N=32768 d=16 RBF, automatic panel widths (2048 / 1024 / 512), scripts/fit_only.py 32768 2, one rocprofv3 --pmc pass per counter group" | head -5
python scripts/summarise_counters.py config2 $O/config2_counters.json "BASELINE configs[2]: N=16384 d=16 Matern-5/2 + cholesky_epsilon, fit x3 + predict(m=1024) x2 + predict_variance x2 (scripts/config_run.py 2), one rocprofv3 --pmc pass per counter group" | head -8
rm -rf $O/*_stats gpurun_out/fit32k_* gpurun_out/config2_*
ls $O
