set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "single_rhs or cached_alpha or solves or predict or posterior" > gpurun_out/r02/pytest2a.log 2>&1; echo "rc=$?" >> gpurun_out/r02/pytest2a.log
tail -25 gpurun_out/r02/pytest2a.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --durations=10 > gpurun_out/r02/pytest2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest2.log
tail -40 gpurun_out/r02/pytest2.log
timeout 600 python scripts/baseline_configs.py > gpurun_out/r02/baseline_trsv.jsonl 2>&1
cat gpurun_out/r02/baseline_trsv.jsonl
