#!/bin/bash
# round 4, GPU run 1: parity after the timing-events / mail changes, the c3 step with and without events, rank 3/8 with a trace, the overlap probe
root=$(pwd); out=$root/gpurun_out/r4_run1; mkdir -p $out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
timeout -k 10 300 python3 bench.py --cpu-sample 0 --second-workload none > $out/bench_c3.json 2> $out/bench_c3.log || { tail -5 $out/bench_c3.log; exit 1; }
python3 - $out/bench_c3.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("c3: %.4f ms/step (events: scan only), %.4f without events; scan kernel %.4f ms frac %.3f" % (j["ms_per_step"], j["ms_per_step_without_timing_events"], j["roofline"]["launch_ms"], j["roofline"]["frac"]))
PY
RANKS="3/8" STEPS=50 tools/shard8.sh r4_run1/shard8 > $out/shard8.log 2>&1 || { tail -5 $out/shard8.log; exit 1; }
grep -E "^rank|step \(|gap" $out/shard8.log | head -40
grep -o '"ms_per_step_without_timing_events": [0-9.]*' $out/shard8/rank_3_of_8.json
timeout -k 10 300 python3 tools/overlap_probe.py c3 20 > $out/overlap_probe.txt 2>&1 || { tail -5 $out/overlap_probe.txt; exit 1; }
cat $out/overlap_probe.txt
