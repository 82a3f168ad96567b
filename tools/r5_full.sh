#!/bin/bash
# the whole -m gpu suite, then the default bench line
root=$(pwd); out=$root/gpurun_out/r5_full; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/tests.txt 2>&1; rc=$?
tail -5 $out/tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5_full/bench.json") if l.startswith("{")][-1])
print("value %.4g  ms_per_step %.4f  without events %.4f" % (d["value"], d["ms_per_step"], d["ms_per_step_without_timing_events"]))
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "launch_ms") if k in d["roofline"]})
print("scan alone", {k: d["roofline_scan"][k] for k in ("achieved", "frac", "launch_ms") if k in d.get("roofline_scan", {})})
print("kernel_ms", d.get("kernel_ms"))
print("conv", d.get("input_conversions_outside_timed_step"))
print("second", (d.get("second_workload") or {}).get("ms_per_step"), (d.get("second_workload") or {}).get("roofline", {}).get("frac"))
PY
