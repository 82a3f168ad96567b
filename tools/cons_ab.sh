#!/bin/bash
# the half-wave consensus kernel in its counting form against the build before it (build/ab/libe2s_hip_seq.so: the sequential per-row loops),
# A/B/A/B on ONE box: per-kernel averages of the bench step under rocprofv3 + the step itself;  tools/cons_ab.sh [bench args, e.g. --workload c3e]
root=$(pwd); out=$root/gpurun_out/r4_cons_ab; mkdir -p $out
for i in 1 2; do
  for v in new seq; do
    if [ $v = seq ]; then export E2S_LIB_PATH=$root/build/ab/libe2s_hip_seq.so; else unset E2S_LIB_PATH; fi
    tools/kstats.sh "$@" > $out/${v}_$i.txt 2>&1 || { tail -5 $out/${v}_$i.txt; exit 1; }
    echo "== $v $i: $(grep -E 'k_consensus' $out/${v}_$i.txt | tr -s ' ') | $(grep -E 'k_flagged' $out/${v}_$i.txt | tr -s ' ' | cut -c1-60) | $(grep 'bench under' $out/${v}_$i.txt)"
  done
done
