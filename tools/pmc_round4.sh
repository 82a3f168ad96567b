#!/bin/bash
# round 4: SQ / TCC counters of the step's kernels over one bench run, one rocprofv3 --pmc pass per counter group (counters only, never with a trace)
#   tools/pmc_round4.sh [workload] [tag]
wl=${1:-c3}; tag=${2:-pmc_r5_$wl}
root=$(pwd); out=$root/gpurun_out/$tag; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --cpu-sample 0 --workload $wl --second-workload none"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_I8 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  E2S_BENCH_SECOND_LEG=1 timeout -k 10 280 rocprofv3 --pmc $grp --kernel-include-regex "^(void )?k_" --output-format csv -d "$out/p$i" -o run -- python3 "$root/bench.py" $ARGS > "$out/p$i.log" 2>&1 || echo "pass $i ($grp) failed: $(tail -n 2 $out/p$i.log | head -c 300)"
done
python3 - "$out" <<'PY' > "$out/summary.txt"
import csv, sys, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("    %-26s mean %.4g  (n=%d, min %.4g, max %.4g)" % (c, sum(v) / len(v), len(v), min(v), max(v)))
PY
find "$out" -name "*.db" -delete
cat "$out/summary.txt" | head -150
