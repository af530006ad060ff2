cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_loss_step_gpu.py tests/test_kernels_gpu.py tests/test_detect_gpu.py -q -k "tower or other_box or test_add or loss_step_vs or forward_train_vs" 2>&1 | tail -15
timeout 300 python bench.py --kind s --batch 512 --no-cpu-baseline --no-exact-bwd --no-other-configs --no-live-traffic > gpurun_out/r05_bench_s512_init.json 2>/dev/null
timeout 400 python tools/make_trained_fixture.py --kind s --iters 2000 --batch 64 --out gpurun_out/yunet_s_synth_trained.pth 2>&1 | tail -4
