#!/bin/bash
# scan workers (= resident waves used) against kernel time, K1 and fused: E2S_SCAN_WORKERS under E2S_DEV=1, two interleaved sweeps on ONE box
root=$(pwd); out=$root/gpurun_out/r4_workers; mkdir -p $out
for rep in 1 2; do
  for w in 7168 3072 4096 5120 2048 6144; do
    E2S_DEV=1 E2S_SCAN_WORKERS=$w timeout -k 10 200 python3 tools/kbench.py c3 > $out/w${w}_$rep.txt 2>&1 || exit 1
    echo "rep $rep workers $w: K1 $(grep -E '^K1    warm' $out/w${w}_$rep.txt | sed 's/.*median \([0-9.]*\) ms.*/\1/') fused $(grep -E '^fused warm' $out/w${w}_$rep.txt | sed 's/.*median \([0-9.]*\) ms.*/\1/')"
  done
done
