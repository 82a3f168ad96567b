#!/bin/bash
# k_cluster_filter2 with dispatcher-fed ranges: parity of the paths that use it, then its time in bench.py's separate-kernels leg at 1 (= one range per wave), 4, 8, 16 rounds
root=$(pwd); out=$root/gpurun_out/r5_filter2; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "prefilter or candidates_equal_oracle or wide_lcp or fused_streaming_adversarial or chunked_prefilter or stale_prefilter" > $out/tests.txt 2>&1; rc=$?
tail -3 $out/tests.txt
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
for r in 1 8 4 16; do
  E2S_DEV=1 E2S_FILTER_ROUNDS=$r timeout -k 10 300 python3 bench.py --workload c3 --steps 10 --warmup 2 --cpu-sample 0 --second-workload none > $out/b_$r.json 2> $out/b_$r.err || { tail -5 $out/b_$r.err; exit 1; }
  python3 - $out/b_$r.json $r <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rp = d.get("roofline_prefilter", {})
print("rounds %2s: k_cluster_filter2 %.4f ms = %.3f of peak   (other mode step %.4f ms; default step %.4f ms)" % (sys.argv[2], rp.get("launch_ms", 0), rp.get("frac", 0), d["other_mode"].get("ms_per_step", 0), d["ms_per_step"]))
PY
done
done
