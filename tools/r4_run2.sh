#!/bin/bash
# round 4, GPU run 2: parity after the job's prepare/enqueue/bail split and the removal of the image-writing scan; the scan at 4 waves per SIMD; counters
root=$(pwd); out=$root/gpurun_out/r4_run2; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
timeout -k 10 200 python3 tools/kbench.py c3 > $out/kbench_default.txt 2>&1 || { tail -5 $out/kbench_default.txt; exit 1; }
E2S_LIB_PATH=$root/build/libe2s_w4.so timeout -k 10 200 python3 tools/kbench.py c3 > $out/kbench_w4.txt 2>&1 || { tail -5 $out/kbench_w4.txt; exit 1; }
echo "== default (3 waves/SIMD fused, 5 scan)"; grep -E "^(K1|fused)" $out/kbench_default.txt
echo "== E2S_SCAN_WAVES=4"; grep -E "^(K1|fused)" $out/kbench_w4.txt
tools/pmc_round4.sh c3 r4_run2/pmc_c3 > $out/pmc.log 2>&1 || true
grep -A40 "k_scan_seg<1, true>" $out/pmc_c3/summary.txt | head -60
