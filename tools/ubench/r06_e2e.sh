cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 900 python tools/train_e2e.py --out gpurun_out/r06_train_e2e.json) > gpurun_out/r06_train_e2e.log 2>&1
grep -v "^Epoch\|amdgpu.ids" gpurun_out/r06_train_e2e.log | tail -20
