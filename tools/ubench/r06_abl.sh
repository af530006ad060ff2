cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${TAG:-r06_bwd64_abl}
: > gpurun_out/$TAG.log
for a in 0 1 2 4 8 16 32 3 12 15 31 47 63; do
  echo "== ABL=$a" >> gpurun_out/$TAG.log
  (ABL=$a ONLY=${ONLY:-80} REPS=300 timeout 120 tools/ubench/bwd_ab.bin libfacedetection.train_amd/libyunet_hip.so libfacedetection.train_amd/libyunet_hip.so) 2>&1 | grep -v "max|\|yardstick" >> gpurun_out/$TAG.log
done
cat gpurun_out/$TAG.log
