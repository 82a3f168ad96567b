#!/bin/bash
# how fast does a page-cached file get into pinned memory on a GPU box: a mapping's first touch against pread, 1 / 4 / 16 threads (tools/io_probe.cpp)
root=$(pwd); out=$root/gpurun_out/r5_io; mkdir -p $out
g++ -O2 -std=c++17 -o /tmp/io_probe tools/io_probe.cpp -lpthread || exit 1
python3 - <<'PY'
import numpy as np
rng = np.random.default_rng(1)
line = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(1 << 20, 151))]
line[:, 150] = 10
with open("/tmp/io_probe.dat", "wb") as f:
    for _ in range(20): f.write(line.tobytes())
PY
ls -l /tmp/io_probe.dat
nproc
/tmp/io_probe /tmp/io_probe.dat > $out/io_probe.txt 2>&1
cat $out/io_probe.txt
rm -f /tmp/io_probe.dat
