#!/bin/bash
# K2b's LDS staging area (E2S_K2B_CAP8, 8-byte units; default 3072 = 24 KB per block of 4 waves: 5 blocks per CU) and grid (E2S_K2B_BLOCKS)
# against the c3 step and one rank of its 8-way cut.  Run through gpurun from the repo root; -> gpurun_out/k2b_knobs.txt
out=gpurun_out/k2b_knobs.txt
: > $out
run() {  # label, env..., -- args
  python3 bench.py "$@" 2>/dev/null | tail -n 1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.4f ms/step' % j['ms_per_step'])"
}
for cfg in "3072 0" "1536 0" "1024 0" "1536 2048" "1536 8192" "3072 2048" "3072 0"; do
  set -- $cfg
  echo -n "c3        cap8 $1 blocks $2: " >> $out
  E2S_K2B_CAP8=$1 E2S_K2B_BLOCKS=$2 run --workload c3 --steps 30 --warmup 5 --cpu-sample 0 --second-workload none >> $out
done
for cfg in "3072 0" "1536 0" "1024 0" "1536 2048" "3072 0"; do
  set -- $cfg
  echo -n "rank 3/8  cap8 $1 blocks $2: " >> $out
  E2S_K2B_CAP8=$1 E2S_K2B_BLOCKS=$2 run --workload c3 --emulate-rank 3/8 --steps 50 --warmup 5 >> $out
done
cat $out
