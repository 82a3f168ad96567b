#!/bin/bash
# Round-5 evidence on the GPU box (through gpurun from the repo root; results under gpurun_out/round5/, copied into profiles/ afterwards):
#   A: tools/profile.sh for c3 and c2 (kernel stats, FETCH_SIZE / WRITE_SIZE passes, clocks) + the c3 timeline
#   B: the default bench line, c3e
part=${1:-A}
root=$(pwd); out=$root/gpurun_out/round5; mkdir -p $out
if [ "$part" = "A" ]; then
  tools/profile.sh round5/c3 c3 > $out/profile_c3.log 2>&1 || { tail -5 $out/profile_c3.log; exit 1; }
  tools/profile.sh round5/c2 c2 > $out/profile_c2.log 2>&1 || { tail -5 $out/profile_c2.log; exit 1; }
  python3 tools/profile_summary.py $out/c3 $out/round5_c3 c3 && python3 tools/profile_summary.py $out/c2 $out/round5_c2 c2
  python3 tools/timeline.py $out/c3/stats 8 > $out/round5_c3_timeline.txt || true
  cat $out/round5_c3_timeline.txt
  find $out -name "*kernel_trace.csv" -size +20M -delete; find $out -name "*.db" -delete
  ls $out
else
  timeout -k 10 900 python3 bench.py > $out/bench.json 2> $out/bench.log || { tail -5 $out/bench.log; exit 1; }
  tail -n 1 $out/bench.json | cut -c1-600
  timeout -k 10 300 python3 bench.py --cpu-sample 0 --second-workload none --workload c3e > $out/bench_c3e.json 2> $out/bench_c3e.log || { tail -5 $out/bench_c3e.log; exit 1; }
  tail -n 1 $out/bench_c3e.json | cut -c1-300
fi
