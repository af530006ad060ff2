cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=libfacedetection.train_amd/libyunet_hip.so
: > gpurun_out/r06_1664_abl.log
for a in 0 1 2 4 8 16 32 15 47 63; do
  echo "== ABL=$a" >> gpurun_out/r06_1664_abl.log
  (ABL=$a SHAPES_ALL=1 REPS=200 timeout 120 tools/ubench/bwd_ab.bin $L $L) 2>&1 | grep -v "max|\|yardstick" | grep "16->64" >> gpurun_out/r06_1664_abl.log
done
awk '/^==/{a=$2} /16->64/{print a, $5, $6}' gpurun_out/r06_1664_abl.log
