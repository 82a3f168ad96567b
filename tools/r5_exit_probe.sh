#!/bin/bash
# what a GPU process pays at exit, piece by piece (tools/exit_probe.cpp)
root=$(pwd); out=$root/gpurun_out/r5_exit; mkdir -p $out
/opt/rocm/bin/hipcc -O2 -o /tmp/exit_probe tools/exit_probe.cpp || exit 1
head -c 6500000000 /dev/zero > /tmp/exit_probe.dat
timeout -k 10 300 /tmp/exit_probe /tmp/exit_probe.dat > $out/exit_probe.txt 2>&1; rc=$?
cat $out/exit_probe.txt; rm -f /tmp/exit_probe.dat; exit $rc
