#!/usr/bin/env python3
"""Prints a slice of a rocprofv3 --kernel-trace (+ --memory-copy-trace) run as a timeline:
start (us, relative), duration, queue, kernel.  usage: timeline.py <dir> <first eval launch> <last>"""
import csv, glob, sys
d, lo_i, hi_i = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:44], r.get("Queue_Id", "")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), ""))
ev.sort()
idx = [i for i, e in enumerate(ev) if "eval_" in e[2]]
lo, hi = idx[lo_i], idx[hi_i]
t0 = ev[lo][0]
for e in ev[lo:hi + 1]:
    print(f"{(e[0] - t0) / 1e3:9.1f} +{(e[1] - e[0]) / 1e3:6.1f} us  q{e[3]:>2} {e[2]}")
