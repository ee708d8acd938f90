#!/usr/bin/env python3
"""Per-forward timeline from a rocprofv3 kernel trace of `bench.py --steps K --warmup W`: one line per forward (stem ...
softmax/value kernel) with its start offset, GPU time, the idle gap in front of it and the mean trunk launch duration.
The last 2*K forwards are the timed region (pipeline 2): prints its GPU-busy fraction.  This is the tool behind the
round-3 diagnosis of the driver's 20-step window (profiles/r03_driver_window.md).

usage: trace_steps.py DIR [timed_forwards]"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
fw, cur = [], None
for s, e, k in ev:
    if "stem15_kernel" in k:
        cur = {"s": s, "e": e, "busy": 0, "trunk": [], "n": 0}
        fw.append(cur)
    if cur is None:
        continue
    cur["e"] = max(cur["e"], e)
    cur["busy"] += e - s
    cur["n"] += 1
    if "trunk15_wino3_kernel" in k:
        cur["trunk"].append(e - s)
    if "head_softmax_value_kernel" in k:
        cur = None
timed = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t0 = fw[0]["s"]
print("forwards %d (last %d = timed region)" % (len(fw), timed))
print("%4s %10s %9s %9s %9s %s" % ("#", "start ms", "gpu us", "gap us", "trunk us", "launches"))
for i, x in enumerate(fw):
    gap = (x["s"] - fw[i - 1]["e"]) / 1e3 if i else 0.0
    tr = sum(x["trunk"]) / max(1, len(x["trunk"])) / 1e3
    mark = " <- timed" if i == len(fw) - timed else ""
    print("%4d %10.3f %9.1f %9.1f %9.2f %d%s" % (i, (x["s"] - t0) / 1e6, x["busy"] / 1e3, gap, tr, x["n"], mark))
tw = fw[-timed:]
span = tw[-1]["e"] - tw[0]["s"]
busy = sum(x["busy"] for x in tw)
trunk = [d for x in tw for d in x["trunk"]]
print("timed region: span %.3f ms, kernel time %.3f ms, busy %.4f, trunk mean %.2f us (first forward %.2f, last %.2f)" % (
    span / 1e6, busy / 1e6, busy / span, sum(trunk) / len(trunk) / 1e3,
    sum(tw[0]["trunk"]) / len(tw[0]["trunk"]) / 1e3, sum(tw[-1]["trunk"]) / len(tw[-1]["trunk"]) / 1e3))
