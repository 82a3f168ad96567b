#!/bin/bash
# A/B of library builds on the scan kernels alone (tools/kbench.py), interleaved on ONE box: tools/ab_kbench.sh <tag> <workload> <copies> <reps> lib1.so lib2.so ...
root=$(pwd); tag=$1; wl=$2; copies=$3; reps=$4; shift 4
out=$root/gpurun_out/$tag; mkdir -p $out
export E2S_DEV=1
for rep in $(seq 1 $reps); do
  for lib in "$@"; do
    name=$(basename $lib .so)
    E2S_LIB_PATH=$root/$lib timeout -k 10 200 python3 tools/kbench.py $wl $copies > $out/${name}_${wl}_${copies}_$rep.txt 2>&1 || { tail -3 $out/${name}_${wl}_${copies}_$rep.txt; exit 1; }
    echo "$wl x$copies rep $rep $name: K1 $(grep -E '^K1    warm' $out/${name}_${wl}_${copies}_$rep.txt | sed 's/.*median \([0-9.]*\) ms.*/\1/') fused $(grep -E '^fused warm' $out/${name}_${wl}_${copies}_$rep.txt | sed 's/.*median \([0-9.]*\) ms.*/\1/')"
  done
done
