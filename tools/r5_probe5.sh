#!/bin/bash
# round 5: where a range's time goes (phase stamps), c3 and one block of it at several range sizes
root=$(pwd); out=$root/gpurun_out/r5_probe5; mkdir -p $out
export E2S_DEV=1 E2S_LIB_PATH=$root/build/libe2s_stamps.so
timeout -k 10 300 python3 tools/scan_stamps.py c3 1 > $out/stamps_c3_fused.txt 2>&1 || exit 1
for rt in 18 9 6 3; do
  E2S_SCAN_RANGE_TILES=$rt timeout -k 10 300 python3 tools/scan_stamps.py c3 1 1 > $out/stamps_blk_fused_rt$rt.txt 2>&1 || exit 1
done
E2S_SCAN_RANGE_TILES=6 timeout -k 10 300 python3 tools/scan_stamps.py c3 0 1 > $out/stamps_blk_k1_rt6.txt 2>&1 || exit 1
for f in $out/*.txt; do echo "== $f"; tail -11 $f; done
