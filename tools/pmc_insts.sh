#!/bin/bash
# development aid: SQ instruction counters of the plane kernels over one bench run (one rocprofv3 --pmc pass, counters only)
#   tools/pmc_insts.sh [workload]
wl=${1:-c2}
root=$(pwd); out=$root/gpurun_out/pmc_insts_$wl; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-include-regex "k_scan_seg|k_cluster_filter2" --output-format csv -d "$out" -o run -- python3 "$root/bench.py" --steps 3 --warmup 1 --cpu-sample 0 --workload $wl > "$out/log.txt" 2>&1
python3 - "$out" <<'PY'
import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in acc.items():
    print(k)
    for c,v in sorted(d.items()): print("   ", c, sum(v)/len(v), len(v))
PY
