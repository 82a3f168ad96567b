#!/bin/bash
# Round-3 evidence on the GPU box (through gpurun from the repo root; results under gpurun_out/round3/, copied into profiles/ afterwards):
#   part A: ranks of the strong cut of c3 on one GPU + timeline; c5s (100 copies) with the chunked job's timing breakdown
#   part B: tools/profile.sh round3_c3 c3 (kernel stats, PMC passes, clocks) and the default bench line
part=${1:-A}
root=$(pwd)
out=$root/gpurun_out/round3
mkdir -p $out
if [ "$part" = "A" ]; then
  STEPS=50 tools/shard8.sh round3/shard8 > $out/shard8.log 2>&1 || { tail -5 $out/shard8.log; exit 1; }
  grep -E "^rank|step \(" $out/shard8.log
  E2S_CLI_TIMING=1 timeout -k 10 500 python3 bench.py --workload c5s > $out/c5s.json 2> $out/c5s.log || { tail -5 $out/c5s.log; exit 1; }
  grep -E "chunked job" $out/c5s.log | tail -n 2
  tail -n 1 $out/c5s.json | cut -c1-200
  E2S_BENCH_C5S_COPIES=30 E2S_CLI_TIMING=1 timeout -k 10 300 python3 bench.py --workload c5s > $out/c5s_30.json 2> $out/c5s_30.log || { tail -5 $out/c5s_30.log; exit 1; }
  grep -E "chunked job" $out/c5s_30.log | tail -n 2
  tail -n 1 $out/c5s_30.json | cut -c1-200
else
  tools/profile.sh round3/c3 c3 > $out/profile.log 2>&1 || { tail -5 $out/profile.log; exit 1; }
  timeout -k 10 900 python3 bench.py > $out/bench.json 2> $out/bench.log || { tail -5 $out/bench.log; exit 1; }
  tail -n 1 $out/bench.json | cut -c1-400
fi
