#!/bin/bash
# k_seg_fix after the change: parity subset, then its average under rocprofv3 on c3 (32768 segments) and on a rank of eight (4096)
root=$(pwd); out=$root/gpurun_out/r5_segfix; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "scan_fuzz or fused_streaming or sharded or few_workers or resident_job or e2c_golden or e2e_golden or drop_in or multi_gpu_driver_on_goldens" > $out/tests.txt 2>&1; rc=$?
tail -3 $out/tests.txt
[ $rc -ne 0 ] && exit $rc
bash tools/kstats.sh --workload c3 --second-workload none 2>&1 | grep -E "k_seg_fix|k_scan_seg|k_job_plan|bench under" | tee $out/c3.txt
bash tools/kstats.sh --workload c3 --second-workload none --emulate-rank 3/8 2>&1 | grep -E "k_seg_fix|k_scan_seg|k_job_plan|bench under" | tee $out/r8.txt
