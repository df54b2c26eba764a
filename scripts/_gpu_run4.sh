cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 300 python scripts/trsv_probe.py > gpurun_out/r02/trsv_probe.txt 2>&1; tail -8 gpurun_out/r02/trsv_probe.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x -k "cholesky or golden or add_rows or substitution or failure or posterior" > gpurun_out/r02/pytest4a.log 2>&1; echo "rc=$?" >> gpurun_out/r02/pytest4a.log
tail -15 gpurun_out/r02/pytest4a.log
for o in 1 0; do timeout 600 python scripts/baseline_configs.py 4096,8192,16384,32768 --panel_fused=$o 2>&1 | cut -c1-200; done > gpurun_out/r02/fused_ab.txt
cat gpurun_out/r02/fused_ab.txt
