cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --durations=8 > gpurun_out/r02/pytest_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest_full.log
tail -25 gpurun_out/r02/pytest_full.log
timeout 600 python scripts/baseline_configs.py > gpurun_out/r02/baseline_now.jsonl 2>&1
cat gpurun_out/r02/baseline_now.jsonl | cut -c1-700
