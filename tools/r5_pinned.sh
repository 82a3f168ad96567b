#!/bin/bash
# uploads out of pageable memory through the handle's pinned buffers (E2S_UPLOAD_PINNED=1, development tier) against the runtime's staging: c5s (211 GB through one GPU), the driver on the c3 files
root=$(pwd); out=$root/gpurun_out/r5_pinned; mkdir -p $out
for v in 0 1; do
  E2S_DEV=1 E2S_UPLOAD_PINNED=$v E2S_CLI_TIMING=1 timeout -k 10 500 python3 bench.py --workload c5s --steps 1 --warmup 0 --cpu-sample 0 > $out/c5s_$v.json 2> $out/c5s_$v.err || { tail -5 $out/c5s_$v.err; exit 1; }
  python3 - $out/c5s_$v.json $v <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("c5s pinned=%s: %.3f s per pass = %.3g entries/s" % (sys.argv[2], d["ms_per_step"] / 1e3, d["value"]))
PY
  grep "chunked job" $out/c5s_$v.err | tail -1 | cut -c1-260
done
E2E_ONLY=mgpu E2E_VARIANTS="pinned:E2S_DEV=1,E2S_UPLOAD_PINNED=1" timeout -k 10 600 python3 tools/e2e_full.py > $out/e2e.txt 2>&1
grep -E "^==|shard uploaded|reads uploaded|ranks done|outside|two copies|from memory" $out/e2e.txt | grep -v "all four" | cut -c1-230 | tail -30
