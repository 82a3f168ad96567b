#!/bin/bash
# Round-4 evidence on the GPU box (through gpurun from the repo root; results under gpurun_out/round4/, copied into profiles/ afterwards):
#   part A: tools/profile.sh for c3 and c2 (kernel stats, FETCH_SIZE / WRITE_SIZE passes, clocks)
#   part B: ranks of the strong cut of c3 on one GPU + timeline, the c3 timeline, the default bench line
#   part C: c5s (100 copies) with the chunked job's timing breakdown, c3e
part=${1:-A}
root=$(pwd); out=$root/gpurun_out/round4; mkdir -p $out
if [ "$part" = "A" ]; then
  tools/profile.sh round4/c3 c3 > $out/profile_c3.log 2>&1 || { tail -5 $out/profile_c3.log; exit 1; }
  tools/profile.sh round4/c2 c2 > $out/profile_c2.log 2>&1 || { tail -5 $out/profile_c2.log; exit 1; }
  python3 tools/profile_summary.py $out/c3 $out/round4_c3 c3 && python3 tools/profile_summary.py $out/c2 $out/round4_c2 c2
  python3 tools/timeline.py $out/c3/stats 16 > $out/round4_c3_timeline.txt || true
  cat $out/round4_c3_timeline.txt
  find $out -name "*kernel_trace.csv" -size +20M -delete; find $out -name "*.db" -delete
elif [ "$part" = "B" ]; then
  STEPS=50 WHICH=16 tools/shard8.sh round4/shard8 > $out/shard8.log 2>&1 || { tail -5 $out/shard8.log; exit 1; }
  grep -E "^rank|step \(" $out/shard8.log
  timeout -k 10 900 python3 bench.py > $out/bench.json 2> $out/bench.log || { tail -5 $out/bench.log; exit 1; }
  tail -n 1 $out/bench.json | cut -c1-400
else
  E2S_CLI_TIMING=1 timeout -k 10 500 python3 bench.py --workload c5s > $out/c5s.json 2> $out/c5s.log || { tail -5 $out/c5s.log; exit 1; }
  grep -E "chunked job" $out/c5s.log | tail -n 2
  tail -n 1 $out/c5s.json | cut -c1-200
  timeout -k 10 300 python3 bench.py --cpu-sample 0 --second-workload none --workload c3e > $out/bench_c3e.json 2> $out/bench_c3e.log || { tail -5 $out/bench_c3e.log; exit 1; }
  tail -n 1 $out/bench_c3e.json | cut -c1-300
fi
