#!/bin/bash
# the layout kernel with the next trip's records prefetched: parity (images), alone (tools/pack_bench.py) and in the step, against head
root=$(pwd); out=$root/gpurun_out/r5_packpf; mkdir -p $out
E2S_LIB_PATH=$root/build/libe2s_packpf.so timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "resident_job or e2c_golden or e2e_golden or sharded or stream_clusters or drop_in or few_workers" > $out/tests.txt 2>&1; rc=$?
tail -1 $out/tests.txt
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do for lib in head packpf; do
  echo "$lib rep $rep: $(E2S_LIB_PATH=$root/build/libe2s_$lib.so timeout -k 10 200 python3 tools/pack_bench.py c3 2>&1 | tail -1)"
done; done
bash tools/ab_libs.sh r5_packpf/ab_c3 c3 3 build/libe2s_head.so build/libe2s_packpf.so || exit 1
AB_EXTRA="--emulate-rank 3/8" bash tools/ab_libs.sh r5_packpf/ab_r8 c3 2 build/libe2s_head.so build/libe2s_packpf.so || exit 1
