cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 2400 python -m pytest -q -m gpu tests/test_modules_autograd_gpu.py tests/test_ddp_gpu.py tests/test_oneshot_gpu.py \
   "tests/test_fullsize_gpu.py::test_full_step_vs_oracle" tests/test_loss_step_gpu.py -s 2>&1 | tail -60) > gpurun_out/r06_call2_tests.log
cat gpurun_out/r06_call2_tests.log
