#!/bin/bash
# the half-wave consensus kernel, A/B/C twice on ONE box: per-kernel averages of the bench step under rocprofv3 + the step itself
#   new    the library as built (left contexts gathered from the reads' bit planes)
#   bytes  the same library with E2S_TAIL_PACKED=0 (development switch: the planes are not built, eight dwords per row from the text)
#   seq    build/ab/libe2s_hip_seq.so, if present: the build before the counting form (sequential per-row loops) -- the library of commit
#          b490c9e: `git worktree add /tmp/seq b490c9e && make -C /tmp/seq lib && cp /tmp/seq/ebwt2snp_amd/libe2s_hip.so build/ab/libe2s_hip_seq.so`
#   tools/cons_ab.sh [bench args, e.g. --workload c3e]
root=$(pwd); out=$root/gpurun_out/r4_cons_ab; mkdir -p $out
vs="new bytes"; [ -z "$NOSEQ" ] && [ -f $root/build/ab/libe2s_hip_seq.so ] && vs="new bytes seq"
for i in 1 2; do
  for v in $vs; do
    unset E2S_LIB_PATH E2S_DEV E2S_TAIL_PACKED
    [ $v = seq ] && export E2S_LIB_PATH=$root/build/ab/libe2s_hip_seq.so
    [ $v = bytes ] && export E2S_DEV=1 E2S_TAIL_PACKED=0
    tools/kstats.sh "$@" > $out/${v}_$i.txt 2>&1 || { tail -5 $out/${v}_$i.txt; exit 1; }
    echo "== $v $i: $(grep -E 'k_consensus' $out/${v}_$i.txt | tr -s ' ') | $(grep -E 'k_flagged' $out/${v}_$i.txt | tr -s ' ' | cut -c1-60) | $(grep 'bench under' $out/${v}_$i.txt)"
  done
done
