#!/usr/bin/env python3
"""Timeline of the last calls in a rocprofv3 --kernel-trace CSV: per kernel its duration and the gap since the previous
kernel's end (what a chain of dependent launches costs beyond its kernels).  usage: trace_gaps.py <dir> [n_last_kernels]"""
import csv, glob, os, re, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda s: re.sub(r"\(.*", "", s.replace("void ", "").replace("nhip::", "").replace("(anonymous namespace)::", ""))[:56]
prev = None
for s, e, k in rows[-n:]:
    print("%-58s dur %8.1f us   gap %8.1f us" % (short(k), (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
