#!/bin/bash
# round 5: the scan with ONE register set for a tile's words (4 waves as it compiles, 5 waves forced) against the two-set kernel: parity subset, A/B
root=$(pwd); out=$root/gpurun_out/r5_oneset; mkdir -p $out
for lib in one4 one5; do
  E2S_LIB_PATH=$root/build/libe2s_$lib.so timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "scan_fuzz or fused_streaming or sharded or few_workers or resident_job or e2c_golden or e2e_golden or wide_lcp" > $out/tests_$lib.txt 2>&1; rc=$?
  echo "$lib: $(tail -1 $out/tests_$lib.txt)"
  [ $rc -ne 0 ] && exit $rc
done
bash tools/ab_kbench.sh r5_oneset/kb c3 10 ${REPS:-3} build/libe2s_head.so build/libe2s_one4.so build/libe2s_one5.so || exit 1
bash tools/ab_libs.sh r5_oneset/ab_c3 c3 ${REPS2:-2} build/libe2s_head.so build/libe2s_one4.so build/libe2s_one5.so || exit 1
AB_EXTRA="--emulate-rank 3/8" bash tools/ab_libs.sh r5_oneset/ab_r8 c3 2 build/libe2s_head.so build/libe2s_one4.so build/libe2s_one5.so || exit 1
