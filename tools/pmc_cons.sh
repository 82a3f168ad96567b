#!/bin/bash
# development aid: counters of the consensus / per-cluster kernels beyond tools/pmc_round4.sh's groups (one rocprofv3 --pmc pass per group, counters only)
#   tools/pmc_cons.sh "<group 1>" "<group 2>" ...
root=$(pwd); out=$root/gpurun_out/pmc_cons; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$out/avail.txt" 2>&1 || true
ARGS="--steps 3 --warmup 1 --cpu-sample 0 --workload c3 --second-workload none"
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $grp --kernel-include-regex "k_consensus|k_flagged|k_tail_write" --output-format csv -d "$out/p$i" -o run -- python3 "$root/bench.py" $ARGS > "$out/p$i.log" 2>&1 || echo "pass $i ($grp) failed: $(grep -i -m2 'error\|invalid\|not' $out/p$i.log | head -c 300)"
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
        print("    %-30s max %.4g  mean %.4g  (n=%d)" % (c, max(v), sum(v) / len(v), len(v)))
PY
find "$out" -name "*.db" -delete
cat "$out/summary.txt"
