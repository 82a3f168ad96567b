#!/bin/bash
# One rank of the N-way STRONG cut of c3 (BASELINE.json configs[3]) on ONE GPU: bench.py --emulate-rank r/N for r = 0, 3, 7 of 8 and one rank
# of the 2- and 4-way cuts; then a rocprofv3 kernel trace of rank 3/8 and its timeline (tools/timeline.py).  Run through gpurun from the
# repo root:  tools/shard8.sh <tag>  ->  gpurun_out/<tag>/
set -e
tag=${1:-shard8}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
STEPS=${STEPS:-50}
for rn in ${RANKS:-0/8 3/8 7/8 1/2 2/4}; do
  f=$out/rank_${rn/\//_of_}
  timeout -k 10 280 python3 bench.py --workload c3 --emulate-rank $rn --steps $STEPS --warmup 5 > $f.json 2> $f.log || { tail -5 $f.log; exit 1; }
  python3 - $f.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("rank %d/%d: %.4f ms/step, %d entries; kernels %s; host phases %s" % (j["emulated_rank"], j["emulated_world"], j["ms_per_step"], j["rank_entries"],
      {k: round(v, 4) for k, v in j["kernel_ms"].items() if not isinstance(v, str)}, j["config"]["host_phase_ms"]))
PY
done
if [ "${TRACE:-1}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o run -- python3 "$root/bench.py" --workload c3 --emulate-rank 3/8 --steps 10 --warmup 2 > "$out/trace.log" 2>&1
  cd "$root"
  python3 tools/timeline.py "$out/trace" ${WHICH:-16} > "$out/timeline_rank3of8.txt" || true
  cat "$out/timeline_rank3of8.txt"
  find "$out/trace" -name "*.db" -delete; find "$out/trace" -name "*kernel_trace.csv" -size +20M -delete
fi
