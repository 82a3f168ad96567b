#!/bin/bash
# development aid: per-kernel averages of one bench run (rocprofv3 --kernel-trace --stats), this repo's kernels only
root=$(pwd); out=$root/gpurun_out/st; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o run -- python3 "$root/bench.py" --steps 10 --warmup 2 --cpu-sample 0 "$@" > "$out/log.txt" 2>&1
rm -f "$out"/*kernel_trace.csv
python3 - "$out/run_kernel_stats.csv" <<'PY'
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('k_scan','k_seg','k_hist','k_cluster','k_flagged','k_ko','k_cand','k_shard','k_apply','k_tile','k_img','k_pack','k_consensus','k_tail','k_phase','k_unpack')):
        print('%-28s calls %s avg %.1f us' % (n.split('(')[0][-28:], r['Calls'], float(r['AverageNs'])/1e3)); tot+=float(r['AverageNs'])/1e3
print('sum of averages %.1f us' % tot)
PY
grep -h '^{' "$out/log.txt" | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('bench under rocprof: ms/step', round(j['ms_per_step'],4), 'value %.3e' % j['value'])"
