#!/bin/bash
# the fused scan with four waves per SIMD against the same code held to three (E2S_SCAN_WORKERS=3072 under E2S_DEV=1), A/B/A/B on ONE box
root=$(pwd); out=$root/gpurun_out/r4_ab_waves_$1; mkdir -p $out
for i in 1 2; do
  timeout -k 10 200 python3 tools/kbench.py c3 > $out/w4_$i.txt 2>&1 || exit 1
  E2S_DEV=1 E2S_SCAN_WORKERS=3072 timeout -k 10 200 python3 tools/kbench.py c3 > $out/w3_$i.txt 2>&1 || exit 1
done
for f in w4_1 w3_1 w4_2 w3_2; do echo "== $f: $(grep -E '^fused warm' $out/$f.txt | cut -c1-75) | $(grep -E '^K1    warm' $out/$f.txt | cut -c40-62)"; done
