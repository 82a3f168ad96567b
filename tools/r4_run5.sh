#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r4_run5; mkdir -p $out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "e2c_golden or scan_fuzz or fused_streaming or e2e_golden or resident_job_images or full_size or sharded or few_workers or wide_lcp" > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
timeout -k 10 200 python3 tools/kbench.py c3 > $out/kbench_default.txt 2>&1 || { tail -5 $out/kbench_default.txt; exit 1; }
E2S_LIB_PATH=$root/build/libe2s_w4.so timeout -k 10 200 python3 tools/kbench.py c3 > $out/kbench_w4.txt 2>&1 || { tail -5 $out/kbench_w4.txt; exit 1; }
timeout -k 10 200 python3 tools/kbench.py c3 > $out/kbench_default2.txt 2>&1 || { tail -5 $out/kbench_default2.txt; exit 1; }
E2S_LIB_PATH=$root/build/libe2s_w4.so timeout -k 10 200 python3 tools/kbench.py c3 > $out/kbench_w4_2.txt 2>&1 || { tail -5 $out/kbench_w4_2.txt; exit 1; }
for f in kbench_default kbench_w4 kbench_default2 kbench_w4_2; do echo "== $f"; grep -E "^(K1|fused)" $out/$f.txt; done
