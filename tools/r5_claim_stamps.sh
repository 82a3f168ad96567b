#!/bin/bash
# where a claimed range's time goes (phase stamps): one block of c3 (3x10^8 entries) at several range sizes, and the whole c3
root=$(pwd); out=$root/gpurun_out/r5_claim_stamps; mkdir -p $out
export E2S_DEV=1 E2S_LIB_PATH=$root/build/libe2s_stamps.so
for rt in 16 4; do
  E2S_SCAN_RANGE_TILES=$rt timeout -k 10 300 python3 tools/scan_stamps.py c3 1 1 > $out/stamps_blk_fused_rt$rt.txt 2>&1 || { tail -5 $out/stamps_blk_fused_rt$rt.txt; exit 1; }
done
timeout -k 10 300 python3 tools/scan_stamps.py c3 1 > $out/stamps_c3_fused.txt 2>&1 || exit 1
for f in $out/*.txt; do echo "== $f"; tail -11 $f; done
