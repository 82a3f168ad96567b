#!/bin/bash
# round 5: instruction diet of the scan: parity subset, then same-box A/B (kernels alone: tools/kbench.py; the step: bench.py)
root=$(pwd); out=$root/gpurun_out/r5_diet; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "${R5_TESTS:-scan_fuzz or fused_streaming or sharded or few_workers or resident_job or e2c_golden or e2e_golden}" > $out/tests.txt 2>&1; rc=$?
tail -3 $out/tests.txt
[ $rc -ne 0 ] && exit $rc
bash tools/ab_kbench.sh r5_diet/kb c3 10 ${REPS:-3} build/libe2s_head.so build/libe2s_diet.so || exit 1
bash tools/ab_libs.sh r5_diet/ab_c3 c3 ${REPS2:-2} build/libe2s_head.so build/libe2s_diet.so || exit 1
