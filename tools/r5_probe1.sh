#!/bin/bash
# round 5, first measurement call: (a) worker start / end stamps of the scan, (b) the overlap probe at 2 and 3 scan workers per SIMD,
# (c) scan workers against time at rank-of-8 size (one c3 block) and at c2 size
root=$(pwd); out=$root/gpurun_out/r5_probe1; mkdir -p $out
export E2S_DEV=1
E2S_LIB_PATH=$root/build/libe2s_stamps.so timeout -k 10 300 python3 tools/scan_stamps.py c3 1 > $out/stamps_c3_fused.txt 2>&1 || { tail -5 $out/stamps_c3_fused.txt; exit 1; }
echo stamps fused done
E2S_LIB_PATH=$root/build/libe2s_stamps.so timeout -k 10 300 python3 tools/scan_stamps.py c3 0 > $out/stamps_c3_k1.txt 2>&1 || exit 1
E2S_LIB_PATH=$root/build/libe2s_stamps.so timeout -k 10 300 python3 tools/scan_stamps.py c3 1 1 > $out/stamps_c3block_fused.txt 2>&1 || exit 1
echo stamps done
for w in 4096 2048 3072; do
  E2S_SCAN_WORKERS=$w timeout -k 10 300 python3 tools/overlap_probe.py c3 20 > $out/overlap_w$w.txt 2>&1 || exit 1
  echo "overlap workers $w: $(tail -1 $out/overlap_w$w.txt)"
done
for rep in 1 2; do
  for w in 4096 1024 2048 3072; do
    E2S_SCAN_WORKERS=$w timeout -k 10 200 python3 tools/kbench.py c3 1 > $out/rank_w${w}_$rep.txt 2>&1 || exit 1
    echo "c3 block rep $rep workers $w: K1 $(grep -E '^K1    warm' $out/rank_w${w}_$rep.txt | sed 's/.*median \([0-9.]*\) ms.*/\1/') fused $(grep -E '^fused warm' $out/rank_w${w}_$rep.txt | sed 's/.*median \([0-9.]*\) ms.*/\1/')"
    E2S_SCAN_WORKERS=$w timeout -k 10 200 python3 tools/kbench.py c2 > $out/c2_w${w}_$rep.txt 2>&1 || exit 1
    echo "c2 rep $rep workers $w: K1 $(grep -E '^K1    warm' $out/c2_w${w}_$rep.txt | sed 's/.*median \([0-9.]*\) ms.*/\1/') fused $(grep -E '^fused warm' $out/c2_w${w}_$rep.txt | sed 's/.*median \([0-9.]*\) ms.*/\1/')"
  done
done
