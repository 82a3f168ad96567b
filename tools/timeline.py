#!/usr/bin/env python3
"""print the kernels / copies of one bench step from a rocprofv3 --kernel-trace (--memory-copy-trace) run, with the idle gaps between them
usage: tools/timeline.py <dir with *_kernel_trace.csv> [which occurrence of the scan kernel, default the 5th]"""
import csv, glob, sys
d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), "q" + r.get("Stream_Id", "?")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", ""), "dma"))
ev.sort()
idx = [i for i, e in enumerate(ev) if e[2].startswith("k_scan_seg<1, true")]
i0, i1 = idx[which], idx[which + 1]
t0 = ev[i0][0]
last_end = t0
for s, e, n, q in ev[i0:i1]:
    gap = (s - last_end) / 1e3
    print("%9.1f %9.1f  %8.1f us  %-4s %-34s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n[:34], ("gap %.1f" % gap) if gap > 3 else ""))
    last_end = max(last_end, e)
print("step (scan start to next scan start): %.1f us" % ((ev[i1][0] - t0) / 1e3))
