cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for tag in ${@:-u1-device-8192 u1-host-8192 u0-host-8192}; do
  rm -rf gpurun_out/pf_$tag
  timeout -k 5 200 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d gpurun_out/pf_$tag -- build/feed_batches 8388608 8 $tag > gpurun_out/pf_$tag.log 2>&1
  tail -1 gpurun_out/pf_$tag.log | cut -c1-250
  f=$(find gpurun_out/pf_$tag -name "*kernel_stats.csv" | head -1); echo "== kernels $tag"; head -12 $f | cut -c1-160
  f=$(find gpurun_out/pf_$tag -name "*hip_api_stats.csv" | head -1); echo "== hip $tag"; head -12 $f | cut -c1-160
  rm -rf gpurun_out/pf_$tag
done
