#!/bin/bash
# round 5: ranges of the scan handed out by the dispatcher -- parity subset, then range sizes against time, then stamps
root=$(pwd); out=$root/gpurun_out/r5_probe2; mkdir -p $out
export E2S_DEV=1
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "e2c_golden_table or scan_fuzz or fused_streaming or sharded or resident_job or few_workers or crafted or candidates_equal or tail_positions" > $out/tests.txt 2>&1 || { tail -30 $out/tests.txt; exit 1; }
tail -3 $out/tests.txt
for rep in 1 2; do
  for rt in 8 4 16 32 1000000; do
    E2S_SCAN_RANGE_TILES=$rt timeout -k 10 200 python3 tools/kbench.py c3 > $out/c3_rt${rt}_$rep.txt 2>&1 || exit 1
    echo "c3 rep $rep range tiles $rt: K1 $(grep -E '^K1    warm' $out/c3_rt${rt}_$rep.txt | sed 's/.*median \([0-9.]*\) ms.*/\1/') fused $(grep -E '^fused warm' $out/c3_rt${rt}_$rep.txt | sed 's/.*median \([0-9.]*\) ms.*/\1/')"
    E2S_SCAN_RANGE_TILES=$rt timeout -k 10 200 python3 tools/kbench.py c3 1 > $out/blk_rt${rt}_$rep.txt 2>&1 || exit 1
    echo "c3 block rep $rep range tiles $rt: K1 $(grep -E '^K1    warm' $out/blk_rt${rt}_$rep.txt | sed 's/.*median \([0-9.]*\) ms.*/\1/') fused $(grep -E '^fused warm' $out/blk_rt${rt}_$rep.txt | sed 's/.*median \([0-9.]*\) ms.*/\1/')"
  done
done
E2S_LIB_PATH=$root/build/libe2s_stamps.so timeout -k 10 300 python3 tools/scan_stamps.py c3 1 > $out/stamps_c3_fused.txt 2>&1 || exit 1
E2S_LIB_PATH=$root/build/libe2s_stamps.so timeout -k 10 300 python3 tools/scan_stamps.py c3 1 1 > $out/stamps_c3block_fused.txt 2>&1 || exit 1
tail -12 $out/stamps_c3_fused.txt
