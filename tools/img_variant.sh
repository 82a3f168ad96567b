#!/bin/bash
# The scan that writes the `.clusters` image itself (k_scan_seg<X,FUSED,IMG=true>, E2S_SCAN_IMG=1: ticketed chunks of 8 tiles, decoupled
# look-back over the chunks' record counts) against the default (6-byte records into per-worker segments + one layout kernel), and its
# ablation builds (E2S_IMG_ABL bits: 1 no image stores, 2 no look-back, 4 no wait for the open-start carry, 8 no histogram; results of
# ablation builds are wrong -- they only time).  Run through gpurun from the repo root: tools/img_variant.sh -> gpurun_out/img_variant.txt
out=gpurun_out/img_variant.txt
: > $out
echo "== default: segments + layout kernel (kbench: scan alone / fused; pack_bench: layout alone)" >> $out
timeout -k 5 200 python3 tools/kbench.py c3 2>&1 | grep -E "warm" >> $out
timeout -k 5 200 python3 tools/pack_bench.py c3 2>&1 | grep -E "^pack" >> $out
echo "== E2S_SCAN_IMG=1: the scan writes the image (no layout kernel)" >> $out
E2S_SCAN_IMG=1 timeout -k 5 200 python3 tools/kbench.py c3 2>&1 | grep -E "warm" >> $out
for v in 2 6 15; do
  echo "== E2S_SCAN_IMG=1, ablation build E2S_IMG_ABL=$v" >> $out
  E2S_SCAN_IMG=1 E2S_LIB_PATH=$PWD/build/libe2s_imgabl$v.so timeout -k 5 200 python3 tools/kbench.py c3 2>&1 | grep -E "warm" >> $out
done
echo "== whole c3 step (bench.py --steps 20): default, then E2S_SCAN_IMG=1 (stepwise calls: the native job does not take this form)" >> $out
timeout -k 5 280 python3 bench.py --workload c3 --steps 20 --warmup 4 --cpu-sample 0 --second-workload none 2>/dev/null | tail -n 1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('default        ms_per_step %.4f  scan kernel %.4f ms' % (j['ms_per_step'], j['roofline']['launch_ms']))" >> $out
E2S_RESIDENT_JOB=0 timeout -k 5 280 python3 bench.py --workload c3 --steps 20 --warmup 4 --cpu-sample 0 --second-workload none 2>/dev/null | tail -n 1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('default, stepwise calls ms_per_step %.4f  scan kernel %.4f ms' % (j['ms_per_step'], j['roofline']['launch_ms']))" >> $out
E2S_SCAN_IMG=1 timeout -k 5 280 python3 bench.py --workload c3 --steps 20 --warmup 4 --cpu-sample 0 --second-workload none 2>/dev/null | tail -n 1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('E2S_SCAN_IMG=1 (stepwise) ms_per_step %.4f  scan kernel %.4f ms' % (j['ms_per_step'], j['roofline']['launch_ms']))" >> $out
cat $out
