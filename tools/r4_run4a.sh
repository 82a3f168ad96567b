#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r4_run4; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
RANKS="3/8" STEPS=50 WHICH=16 tools/shard8.sh r4_run4/shard8 > $out/shard8.log 2>&1 || { tail -5 $out/shard8.log; exit 1; }
grep -E "^rank" $out/shard8.log | head; cat $out/shard8/timeline_rank3of8.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_c3" -o run -- python3 "$root/bench.py" --steps 10 --warmup 2 --cpu-sample 0 --second-workload none > "$out/trace_c3.log" 2>&1
cd "$root"; python3 tools/timeline.py "$out/trace_c3" 16 > "$out/timeline_c3.txt" || true; cat "$out/timeline_c3.txt"
grep -h '^{' $out/trace_c3.log | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('c3 under rocprof: %.4f ms/step, %.4f without events' % (j['ms_per_step'], j['ms_per_step_without_timing_events']))"
find "$out" -name "*.db" -delete; find "$out" -name "*kernel_trace.csv" -size +20M -delete
