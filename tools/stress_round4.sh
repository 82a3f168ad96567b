#!/bin/bash
# round 4: random cases beside the reference on the round's final kernels (CLI seeds, half of the second batch with ragged reads; random columns)
root=$(pwd); out=$root/gpurun_out/r4_stress; mkdir -p $out
timeout -k 10 420 python3 tools/stress_cli.py 7000 7200 > $out/cli.txt 2>&1; echo "cli rc $?" >> $out/cli.txt; tail -2 $out/cli.txt
E2S_STRESS_RAGGED=1 timeout -k 10 330 python3 tools/stress_cli.py 7200 7350 > $out/cli_ragged.txt 2>&1; echo "cli ragged rc $?" >> $out/cli_ragged.txt; tail -2 $out/cli_ragged.txt
E2S_DEV=1 timeout -k 10 330 python3 tools/stress_columns.py 8000 8300 > $out/columns.txt 2>&1; echo "columns rc $?" >> $out/columns.txt; tail -2 $out/columns.txt
