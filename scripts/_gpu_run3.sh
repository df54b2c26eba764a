cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 300 python scripts/cond_debug.py > gpurun_out/r02/cond_debug.txt 2>&1
cat gpurun_out/r02/cond_debug.txt
for pad in 0 128 256 512; do timeout 300 python scripts/baseline_configs.py 32768 --ld_pad=$pad 2>&1 | cut -c1-420; done > gpurun_out/r02/ldpad.txt
cat gpurun_out/r02/ldpad.txt
