#!/bin/bash
# round 5: the scan as one-wave blocks -- range sizes against time on c3, on one block of it (a rank of eight) and on c2
root=$(pwd); out=$root/gpurun_out/r5_probe4; mkdir -p $out
export E2S_DEV=1
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "e2c_golden_table or scan_fuzz or fused_streaming_fuzz or sharded_fuzz or resident_job_images" > $out/tests.txt 2>&1 || { tail -30 $out/tests.txt; exit 1; }
tail -2 $out/tests.txt
for rep in 1 2; do
  for rt in 8 32 1000000; do
    E2S_SCAN_RANGE_TILES=$rt timeout -k 10 200 python3 tools/kbench.py c3 > $out/c3_rt${rt}_$rep.txt 2>&1 || exit 1
    echo "c3 rep $rep range tiles $rt: K1 $(grep -E '^K1    warm' $out/c3_rt${rt}_$rep.txt | sed 's/.*median \([0-9.]*\) ms.*/\1/') fused $(grep -E '^fused warm' $out/c3_rt${rt}_$rep.txt | sed 's/.*median \([0-9.]*\) ms.*/\1/')"
  done
  for rt in 18 9 6 5 4 3; do
    E2S_SCAN_RANGE_TILES=$rt timeout -k 10 200 python3 tools/kbench.py c3 1 > $out/blk_rt${rt}_$rep.txt 2>&1 || exit 1
    echo "c3 block rep $rep range tiles $rt: K1 $(grep -E '^K1    warm' $out/blk_rt${rt}_$rep.txt | sed 's/.*median \([0-9.]*\) ms.*/\1/') fused $(grep -E '^fused warm' $out/blk_rt${rt}_$rep.txt | sed 's/.*median \([0-9.]*\) ms.*/\1/')"
  done
done
E2S_LIB_PATH=$root/build/libe2s_stamps.so timeout -k 10 300 python3 tools/scan_stamps.py c3 1 > $out/stamps_c3_fused.txt 2>&1 || exit 1
E2S_SCAN_RANGE_TILES=6 E2S_LIB_PATH=$root/build/libe2s_stamps.so timeout -k 10 300 python3 tools/scan_stamps.py c3 1 1 > $out/stamps_c3block_fused_rt6.txt 2>&1 || exit 1
tail -9 $out/stamps_c3_fused.txt; tail -9 $out/stamps_c3block_fused_rt6.txt
