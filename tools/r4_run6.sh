#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r4_run6; mkdir -p $out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "e2c_golden or scan_fuzz or fused_streaming or e2e_golden or resident_job_images or full_size or sharded or few_workers or wide_lcp or multiples" > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
E2S_LIB_PATH=$root/build/libe2s_unpinned.so timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "scan_fuzz or fused_streaming or e2e_golden or full_size" > $out/pytest_unpinned.log 2>&1 || { tail -40 $out/pytest_unpinned.log; exit 1; }
tail -2 $out/pytest_unpinned.log
for i in 1 2; do
timeout -k 10 200 python3 tools/kbench.py c3 > $out/kbench_default_$i.txt 2>&1 || { tail -5 $out/kbench_default_$i.txt; exit 1; }
E2S_LIB_PATH=$root/build/libe2s_unpinned.so timeout -k 10 200 python3 tools/kbench.py c3 > $out/kbench_unpinned_$i.txt 2>&1 || { tail -5 $out/kbench_unpinned_$i.txt; exit 1; }
done
for f in kbench_default_1 kbench_unpinned_1 kbench_default_2 kbench_unpinned_2; do echo "== $f"; grep -E "^(K1|fused)" $out/$f.txt; done
