#!/bin/bash
# kernel trace of a few c3 steps -> timeline of one step + kernel statistics
root=$(pwd); out=$root/gpurun_out/${1:-r5_timeline}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 $root/bench.py --steps 10 --warmup 2 --cpu-sample 0 --workload ${2:-c3} --second-workload none > $out/stats.log 2>&1 || { tail -5 $out/stats.log; exit 1; }
cd $root
python3 tools/timeline.py $out/stats 8 > $out/timeline.txt; cat $out/timeline.txt
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); head -25 $f | cut -d, -f1-8 | column -s, -t | cut -c1-200
grep -h '^{' $out/stats.log | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('under rocprof: ms_per_step', d['ms_per_step'], 'scan', d['roofline']['launch_ms'])"
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -size +20M -delete
