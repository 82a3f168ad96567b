#!/bin/bash
# round 5: the scan with claimed ranges against the round's dispatcher-fed form: parity subset, then same-box A/B on c3 and on a rank of eight
root=$(pwd); out=$root/gpurun_out/r5_claim; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "scan_fuzz or fused_streaming or sharded or few_workers or resident_job or e2c_golden or e2e_golden or wide_lcp or tiled_copies or beyond_2_pow_32" > $out/tests.txt 2>&1; rc=$?
tail -3 $out/tests.txt
[ $rc -ne 0 ] && exit $rc
bash tools/ab_libs.sh r5_claim/ab_c3 c3 ${REPS:-2} build/libe2s_head.so build/libe2s_claim.so || exit 1
for rt in ${RTS:-8 4}; do
  echo "== E2S_SCAN_RANGE_TILES=$rt =="
  E2S_DEV=1 E2S_SCAN_RANGE_TILES=$rt bash tools/ab_libs.sh r5_claim/ab_c3_rt$rt c3 1 build/libe2s_claim.so || exit 1
done
echo "== rank 3 of 8 =="
AB_EXTRA="--emulate-rank 3/8" bash tools/ab_libs.sh r5_claim/ab_r8 c3 ${REPS:-2} build/libe2s_head.so build/libe2s_claim.so || exit 1
for rt in ${RTS8:-8 4 2}; do
  echo "== rank 3 of 8, E2S_SCAN_RANGE_TILES=$rt =="
  E2S_DEV=1 E2S_SCAN_RANGE_TILES=$rt AB_EXTRA="--emulate-rank 3/8" bash tools/ab_libs.sh r5_claim/ab_r8_rt$rt c3 1 build/libe2s_claim.so || exit 1
done
