#!/bin/bash
# END / START bits walked together (-DE2S_WALK_PAIRED) against the four separate loops: parity subset on the variant build, then A/B
root=$(pwd); out=$root/gpurun_out/r5_paired; mkdir -p $out
E2S_LIB_PATH=$root/build/libe2s_paired.so timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "scan_fuzz or fused_streaming or sharded or few_workers or resident_job or e2c_golden or e2e_golden" > $out/tests.txt 2>&1; rc=$?
tail -1 $out/tests.txt
[ $rc -ne 0 ] && exit $rc
bash tools/ab_kbench.sh r5_paired/kb c3 10 ${REPS:-3} build/libe2s_head.so build/libe2s_paired.so || exit 1
AB_EXTRA="--emulate-rank 3/8" bash tools/ab_libs.sh r5_paired/ab_r8 c3 2 build/libe2s_head.so build/libe2s_paired.so || exit 1
