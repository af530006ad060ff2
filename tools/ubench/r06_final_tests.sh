cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 3000 python -m pytest tests -q -m gpu --durations=15 2>&1 | tail -45) > gpurun_out/r06_pytest_gpu.log
cat gpurun_out/r06_pytest_gpu.log
