#!/bin/bash
# PMC passes over chosen kernels of the bench step (run through gpurun from the repo root):
#   tools/pmc_kernels.sh <tag> <kernel regex> [workload]  -> gpurun_out/<tag>/<pass>/...
# every pass is its own rocprofv3 run with --pmc only (never combined with a trace), python3 the direct child.
set -e
tag=${1:-pmc}
rx=${2:-k_consensus}
wl=${3:-c3}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
  "FETCH_SIZE TCC_HIT_sum" \
  "TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
  "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-include-regex "$rx" --output-format csv -d "$out/p$i" -o run -- python3 "$root/bench.py" --steps 2 --warmup 1 --cpu-sample 0 --workload $wl --second-workload none > "$out/p$i.log" 2>&1 || echo "pass $i failed"
  tail -2 "$out/p$i.log" | cut -c1-300
done
find "$out" -name "*.db" -delete
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
