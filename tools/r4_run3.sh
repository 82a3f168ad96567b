#!/bin/bash
# round 4, GPU run 3: parity (consensus loops, mgpu 2 B/entry form), c3 bench, rank 3/8 + trace
root=$(pwd); out=$root/gpurun_out/r4_run3; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
timeout -k 10 300 python3 bench.py --cpu-sample 0 --second-workload none > $out/bench_c3.json 2> $out/bench_c3.log || { tail -5 $out/bench_c3.log; exit 1; }
python3 - $out/bench_c3.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("c3: %.4f ms/step (events: scan only), %.4f without events; scan kernel %.4f ms frac %.3f" % (j["ms_per_step"], j["ms_per_step_without_timing_events"], j["roofline"]["launch_ms"], j["roofline"]["frac"]))
PY
RANKS="3/8" STEPS=50 tools/shard8.sh r4_run3/shard8 > $out/shard8.log 2>&1 || { tail -5 $out/shard8.log; exit 1; }
grep -E "^rank|step \(" $out/shard8.log | head; cat $out/shard8/timeline_rank3of8.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_c3" -o run -- python3 "$root/bench.py" --steps 10 --warmup 2 --cpu-sample 0 --second-workload none > "$out/trace_c3.log" 2>&1
cd "$root"; python3 tools/timeline.py "$out/trace_c3" 16 > "$out/timeline_c3.txt" || true; cat "$out/timeline_c3.txt"
find "$out" -name "*.db" -delete; find "$out" -name "*kernel_trace.csv" -size +20M -delete
