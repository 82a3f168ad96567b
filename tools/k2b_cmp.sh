cd /tmp && export TMPDIR=/tmp
for v in new old; do
  if [ $v = old ]; then export E2S_LIB_PATH=$GRAFT_REPO_ROOT/build/libe2s_oldk2b.so; else unset E2S_LIB_PATH; fi
  for at in 2 0; do
    rm -rf $GRAFT_REPO_ROOT/gpurun_out/k2bt; E2S_JOB_IMAGE_AT=$at rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/k2bt -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 6 --warmup 2 --cpu-sample 0 --second-workload none > /dev/null 2>&1
    echo "== $v kernel, image at $at"; python3 $GRAFT_REPO_ROOT/tools/timeline.py $GRAFT_REPO_ROOT/gpurun_out/k2bt 5 | grep -E "flagged|pack_seg|ko_scan|consensus|step"
  done
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/k2bt
