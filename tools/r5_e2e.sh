#!/bin/bash
# round 5: the one-process driver on the whole c3 files (tools/e2e_full.py, E2E_ONLY=mgpu) under variants: tools/r5_e2e.sh "<E2E_VARIANTS>" [tag]
root=$(pwd); out=$root/gpurun_out/r5_e2e; mkdir -p $out; tag=${2:-run}
E2E_ONLY=mgpu E2E_VARIANTS="$1" timeout -k 10 900 python3 tools/e2e_full.py > $out/e2e_$tag.txt 2>&1; rc=$?
grep -E "^==|summary|timing|outside|upload\]" $out/e2e_$tag.txt | grep -v "index files mapped\|output files closed\|step done" | cut -c1-260 | tail -120
exit $rc
