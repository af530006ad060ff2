cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -q -k "sgd or (dp_bwd and 32) or fused_pooling or (full_step and s-320)" 2>&1 | tail -6
for v in 0 1; do
  echo "== YUNET_BWD32_SPLIT=$v" >> gpurun_out/r05_bench_s512_ab.log
  YUNET_BWD32_SPLIT=$v timeout 300 python bench.py --kind s --batch 512 --steps 40 --warmup 10 --no-cpu-baseline --no-exact-bwd --no-other-configs --no-live-traffic 2>/dev/null | tail -1 | python -c "
import json,sys
b=json.loads(sys.stdin.read()); print(b['ms_per_step'], b['value'], b['weights'])
for k,v in list(b['kernels'].items())[:6]: print('   ',k,v)
" >> gpurun_out/r05_bench_s512_ab.log 2>&1
done
cat gpurun_out/r05_bench_s512_ab.log
