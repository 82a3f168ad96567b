#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r4_run4; mkdir -p $out
timeout -k 10 300 python3 bench.py --cpu-sample 0 --second-workload none > $out/bench_c3.json 2> $out/bench_c3.log || { tail -5 $out/bench_c3.log; exit 1; }
timeout -k 10 300 python3 bench.py --cpu-sample 0 --second-workload none --workload c3e > $out/bench_c3e.json 2> $out/bench_c3e.log || { tail -5 $out/bench_c3e.log; exit 1; }
python3 - $out/bench_c3.json $out/bench_c3e.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    c = j["config"]
    print("%s: %.4f ms/step (%.4f without events); scan %.4f ms frac %.3f; records %d, flagged fraction %.5f, candidates %d, events %d, native %s, mcl %d" % (
        c["workload"][:4], j["ms_per_step"], j["ms_per_step_without_timing_events"], j["roofline"]["launch_ms"], j["roofline"]["frac"], c["cluster_records"],
        c["flagged_fraction_of_records"], c["candidates"], c["snp_events"], c["one_enqueue_one_wait"], c["max_clust_length"]))
PY
timeout -k 10 800 python3 tools/e2e_full.py 10 > $out/e2e_full.txt 2>&1 || { tail -30 $out/e2e_full.txt; exit 1; }
grep -E "^==|summary|^   (two|ebwt2snp)" $out/e2e_full.txt
