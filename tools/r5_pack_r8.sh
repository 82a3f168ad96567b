#!/bin/bash
# a rank of eight: the layout kernel's block count (E2S_PACK_BLOCKS, development tier) in the step
root=$(pwd); out=$root/gpurun_out/r5_pack_r8; mkdir -p $out
for rep in 1 2; do
for pb in 2048 4096 1024; do
  E2S_DEV=1 E2S_PACK_BLOCKS=$pb E2S_BENCH_PHASES=1 E2S_BENCH_SECOND_LEG=0 timeout -k 10 300 python3 bench.py --workload c3 --steps 20 --warmup 3 --cpu-sample 0 --second-workload none --emulate-rank 3/8 > $out/b_$pb.json 2> $out/b_$pb.err || { tail -5 $out/b_$pb.err; exit 1; }
  python3 - $out/b_$pb.json $pb <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
k = d.get("kernel_ms") or {}
print("pack blocks %5s: step %.4f ms (no events %.4f)  scan %.4f  ->K2b end %.4f  tail %.4f" % (sys.argv[2], d["ms_per_step"], d["ms_per_step_without_timing_events"], d["roofline"]["launch_ms"], k.get("scan_start_to_K2b_end", 0), k.get("tail", 0)))
PY
done
done
