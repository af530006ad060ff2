cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest -q -m gpu tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_bf16_gpu.py 2>&1 | tail -5
