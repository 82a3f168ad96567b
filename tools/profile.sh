#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   tools/profile.sh <tag> [workload] -> gpurun_out/<tag>/{stats,FETCH_SIZE,WRITE_SIZE}/...
# kernel-trace/stats and each PMC counter are separate passes (never combined), the program is the
# direct child of rocprofv3.  Summarise afterwards with tools/profile_summary.py into profiles/.
set -e
tag=${1:-prof}
wl=${2:-c2}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --cpu-sample 0 --workload $wl --second-workload none"
# the box's clocks while the bench runs (sampled beside it; VERDICT r1: a fraction of peak needs the sustained clock next to it)
( while true; do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk"; sleep 0.5; done > "$out/clocks_under_load.txt" ) &
SAMPLER=$!
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- python3 "$root/bench.py" $ARGS > "$out/stats.log" 2>&1
kill $SAMPLER 2>/dev/null || true
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-include-regex "^(void )?k_" --output-format csv -d "$out/$c" -o run -- python3 "$root/bench.py" --steps 3 --warmup 1 --cpu-sample 0 --workload $wl --second-workload none > "$out/$c.log" 2>&1
done
find "$out" -name "*.db" -delete; find "$out" -name "*kernel_trace.csv" -size +20M -delete
du -sh "$out" "$out"/* | tail -8
grep -h '^{' "$out/stats.log" > "$out/bench_under_rocprof.json" || true
echo done
