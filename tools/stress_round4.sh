#!/bin/bash
# round 4: random cases beside the reference on the round's final kernels: tools/stress_round4.sh <first CLI seed> <CLI seeds> <first column seed> <column seeds> [<first seed> <seeds> of CLI cases with sequencing errors]
# (half of the CLI seeds with ragged reads); output under gpurun_out/r4_stress/
root=$(pwd); out=$root/gpurun_out/r4_stress; mkdir -p $out
c0=${1:-7000}; cn=${2:-200}; k0=${3:-8000}; kn=${4:-300}; c40=${5:-9000}; c4n=${6:-}
timeout -k 10 500 python3 tools/stress_cli.py $c0 $((c0 + cn / 2)) > $out/cli_$c0.txt 2>&1; echo "cli rc $?" >> $out/cli_$c0.txt; tail -2 $out/cli_$c0.txt
E2S_STRESS_RAGGED=1 timeout -k 10 500 python3 tools/stress_cli.py $((c0 + cn / 2)) $((c0 + cn)) > $out/cli_ragged_$c0.txt 2>&1; echo "cli ragged rc $?" >> $out/cli_ragged_$c0.txt; tail -2 $out/cli_ragged_$c0.txt
[ -n "$c4n" ] && { E2S_STRESS_ERRORS=1 timeout -k 10 500 python3 tools/stress_cli.py $c40 $((c40 + c4n)) > $out/cli_errors_$c40.txt 2>&1; echo "cli errors rc $?" >> $out/cli_errors_$c40.txt; tail -2 $out/cli_errors_$c40.txt; }
E2S_DEV=1 timeout -k 10 500 python3 tools/stress_columns.py $k0 $((k0 + kn)) > $out/columns_$k0.txt 2>&1; echo "columns rc $?" >> $out/columns_$k0.txt; tail -2 $out/columns_$k0.txt
