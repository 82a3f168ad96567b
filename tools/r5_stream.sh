#!/bin/bash
# round 5: the uploads from files (pread into pinned buffers) and the streamed FASTA: their tests, the I/O probe, the whole c3 files end to end
root=$(pwd); out=$root/gpurun_out/r5_stream; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "${R5_TESTS:-streamed_fasta or uploads_from_files or multi_gpu_driver_on_goldens or drop_in or chunked_executables or ragged}" > $out/tests.txt 2>&1; rc=$?
tail -5 $out/tests.txt
[ $rc -ne 0 ] && exit $rc
bash tools/r5_io_probe.sh > $out/io_probe_call.txt 2>&1; tail -30 $out/io_probe_call.txt
E2E_ONLY=mgpu E2E_VARIANTS="${E2E_VARIANTS:-pread:E2S_MGPU_PREAD=1;stream:E2S_MGPU_STREAM_READS=1;both:E2S_MGPU_PREAD=1,E2S_MGPU_STREAM_READS=1}" timeout -k 10 900 python3 tools/e2e_full.py > $out/e2e.txt 2>&1; rc=$?
grep -E "^==|summary|^   ebwt2snp|timing|outside|upload\]" $out/e2e.txt | cut -c1-220 | tail -90
exit $rc
