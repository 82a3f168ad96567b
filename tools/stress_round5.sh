#!/bin/bash
# round 5: random cases beside the reference / the oracle on the round's final kernels -- the scan's ranges in three shapes (a tile per range: the
# default for inputs this small; ~10 tiles; 3 tiles) so that the look-back at a range's start meets every kind of boundary; output under gpurun_out/r5_stress/
#   tools/stress_round5.sh <first CLI seed> <CLI seeds> <first column seed> <column seeds>
root=$(pwd); out=$root/gpurun_out/r5_stress; mkdir -p $out
c0=${1:-17000}; cn=${2:-200}; k0=${3:-18000}; kn=${4:-300}
timeout -k 10 400 python3 tools/stress_cli.py $c0 $((c0 + cn / 2)) > $out/cli_$c0.txt 2>&1; echo "cli rc $?" >> $out/cli_$c0.txt; tail -2 $out/cli_$c0.txt
E2S_STRESS_RAGGED=1 E2S_STRESS_ERRORS=1 timeout -k 10 400 python3 tools/stress_cli.py $((c0 + cn / 2)) $((c0 + cn)) > $out/cli_ragged_errors_$c0.txt 2>&1; echo "cli ragged+errors rc $?" >> $out/cli_ragged_errors_$c0.txt; tail -2 $out/cli_ragged_errors_$c0.txt
E2S_DEV=1 timeout -k 10 400 python3 tools/stress_columns.py $k0 $((k0 + kn / 3)) > $out/columns_$k0.txt 2>&1; echo "columns rc $?" >> $out/columns_$k0.txt; tail -2 $out/columns_$k0.txt
E2S_DEV=1 E2S_SCAN_WORKERS=7 E2S_SCAN_RANGE_TILES=10 E2S_FILTER_WORKERS=5 timeout -k 10 400 python3 tools/stress_columns.py $((k0 + kn / 3)) $((k0 + 2 * kn / 3)) > $out/columns_10tiles_$k0.txt 2>&1; echo "columns (10-tile ranges) rc $?" >> $out/columns_10tiles_$k0.txt; tail -2 $out/columns_10tiles_$k0.txt
E2S_DEV=1 E2S_SCAN_WORKERS=20 E2S_SCAN_RANGE_TILES=3 timeout -k 10 400 python3 tools/stress_columns.py $((k0 + 2 * kn / 3)) $((k0 + kn)) > $out/columns_3tiles_$k0.txt 2>&1; echo "columns (3-tile ranges) rc $?" >> $out/columns_3tiles_$k0.txt; tail -2 $out/columns_3tiles_$k0.txt
