#!/bin/bash
# round 5: long ranges first, short ones at the end; the carries' loads in one batch -- parity subset, then the schedule's parameters against time
root=$(pwd); out=$root/gpurun_out/r5_probe6; mkdir -p $out
export E2S_DEV=1
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "e2c_golden_table or scan_fuzz or fused_streaming or sharded or resident_job_images or few_workers or crafted or candidates_equal" > $out/tests.txt 2>&1 || { tail -30 $out/tests.txt; exit 1; }
tail -2 $out/tests.txt
run() { # workload copies ts factor rounds
  E2S_SCAN_RANGE_TILES=$3 E2S_SCAN_BIG_FACTOR=$4 E2S_SCAN_SMALL_ROUNDS=$5 timeout -k 10 200 python3 tools/kbench.py $1 $2 > $out/k_$1_$2_$3_$4_$5_$6.txt 2>&1 || exit 1
  echo "$1 x$2 rep $6 ts $3 factor $4 rounds $5: K1 $(grep -E '^K1    warm' $out/k_$1_$2_$3_$4_$5_$6.txt | sed 's/.*median \([0-9.]*\) ms.*/\1/') fused $(grep -E '^fused warm' $out/k_$1_$2_$3_$4_$5_$6.txt | sed 's/.*median \([0-9.]*\) ms.*/\1/')"
}
for rep in 1 2; do
  run c3 10 8 4 3 $rep; run c3 10 8 4 1 $rep; run c3 10 12 2 2 $rep; run c3 10 6 6 3 $rep; run c3 10 23 1 1 $rep
  run c3 1 2 4 3 $rep; run c3 1 3 4 2 $rep; run c3 1 4 3 1 $rep; run c3 1 3 3 2 $rep; run c3 1 6 2 1 $rep; run c3 1 18 1 1 $rep; run c3 1 2 8 2 $rep
  run c2 1 1 4 3 $rep; run c2 1 2 4 2 $rep; run c2 1 3 3 1 $rep; run c2 1 12 1 1 $rep
done
export E2S_LIB_PATH=$root/build/libe2s_stamps.so
timeout -k 10 300 python3 tools/scan_stamps.py c3 1 > $out/stamps_c3_fused.txt 2>&1 || exit 1
timeout -k 10 300 python3 tools/scan_stamps.py c3 1 1 > $out/stamps_blk_fused.txt 2>&1 || exit 1
tail -11 $out/stamps_c3_fused.txt; tail -11 $out/stamps_blk_fused.txt
