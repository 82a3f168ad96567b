#!/bin/bash
# A/B of whole library builds on ONE box, interleaved: tools/ab_libs.sh <out tag> <workload> <reps> lib1.so lib2.so ...   (bench.py's step, phases by events in a leg of their own)
root=$(pwd); tag=$1; wl=$2; reps=$3; shift 3
out=$root/gpurun_out/$tag; mkdir -p $out
for rep in $(seq 1 $reps); do
  for lib in "$@"; do
    name=$(basename $lib .so)
    E2S_LIB_PATH=$root/$lib E2S_BENCH_PHASES=1 E2S_BENCH_SECOND_LEG=0 timeout -k 10 300 python3 bench.py --workload $wl --steps 20 --warmup 3 --cpu-sample 0 --second-workload none $AB_EXTRA > $out/${name}_$rep.json 2> $out/${name}_$rep.err || { tail -5 $out/${name}_$rep.err; exit 1; }
    python3 - $out/${name}_$rep.json $name $rep <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
k = d.get("kernel_ms") or {}
print("%-16s rep %s: step %.4f ms (no events %.4f)  scan kernel %.4f  scan..fix %.4f  ->K2b end %.4f  tail %.4f" % (
    sys.argv[2], sys.argv[3], d["ms_per_step"], d["ms_per_step_without_timing_events"], d["roofline"]["launch_ms"], k.get("scan+prefilter+fix+stats", 0),
    k.get("scan_start_to_K2b_end", 0), k.get("tail", 0)))
PY
  done
done
