#!/bin/bash
# development aid: SQ instruction counters of the scan kernel (tools/kbench.py), one rocprofv3 --pmc pass
root=$(pwd); out=$root/gpurun_out/pmc_k1; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d "$out" -o run -- python3 "$root/tools/kbench.py" > "$out/log.txt" 2>&1
python3 - "$out" <<'PY'
import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_scan_seg' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k, sum(v)/len(v), len(v))
PY
