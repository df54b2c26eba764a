# When GPU use is reopened: validate HEAD and refresh the round's profiles in ONE bounded call (every part under its own time limit).
#   gpurun --timeout 2400 -- 'bash scripts/gpu_back.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06p
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r06p/gpu_suite.txt; cat gpurun_out/r06p/gpu_suite.txt
timeout 900 bash scripts/profile_r06.sh > gpurun_out/r06p/profile_log.txt 2>&1; tail -5 gpurun_out/r06p/profile_log.txt
AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3 timeout 500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_rccl_multi.py 2>&1 | tail -3 > gpurun_out/r06p/serialized_suite.txt; cat gpurun_out/r06p/serialized_suite.txt
