set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
(nproc; lscpu | grep -i "model name\|^CPU(s)\|Thread\|Socket"; free -g | head -2) > gpurun_out/r02/host.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r02/pytest1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest1.log
tail -40 gpurun_out/r02/pytest1.log
timeout 600 python scripts/baseline_configs.py > gpurun_out/r02/baseline_before.jsonl 2>&1
cat gpurun_out/r02/baseline_before.jsonl
