# round 6, call 1: the new parity / drop-in / two-rank tests, the one-shot probe, a bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 1500 python -m pytest -x -q -m gpu tests/test_modules_autograd_gpu.py tests/test_ddp_gpu.py tests/test_oneshot_gpu.py \
   "tests/test_fullsize_gpu.py::test_full_step_vs_oracle" tests/test_loss_step_gpu.py -s 2>&1 | tail -40) > gpurun_out/r06_call1_tests.log
cat gpurun_out/r06_call1_tests.log
(timeout 300 python tools/oneshot_probe.py) > gpurun_out/r06_oneshot_probe.json 2> gpurun_out/r06_oneshot_probe.err; cat gpurun_out/r06_oneshot_probe.json; tail -3 gpurun_out/r06_oneshot_probe.err
(timeout 600 python bench.py --no-other-configs) > gpurun_out/r06_bench_call1.json 2> gpurun_out/r06_bench_call1.err; tail -c 3000 gpurun_out/r06_bench_call1.json
