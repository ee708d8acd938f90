#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel-trace stats + PMC passes) into a small markdown table.

FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3); on gfx950 FETCH_SIZE counts 64 B per 128-B request
for wide coalesced reads, so read bytes are reported both raw and x2 (MI355X_MICROARCH.md, HBM)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def short(name):
    name = name.replace("void apz::", "").replace("apz::", "")
    return name[:70]


def main():
    root = sys.argv[1]
    print("# rocprofv3 summary (%s)\n" % os.path.basename(os.path.abspath(root)))
    # ---- kernel trace: per-kernel count / total / avg duration
    rows = defaultdict(list)
    for f in find(os.path.join(root, "trace"), "*kernel_trace.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in rows.values()) or 1.0
    print("## kernel trace (`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps N --warmup 10 --no-extras`)\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % GPU time |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        print("| `%s` | %d | %.2f | %.1f | %.1f | %.1f | %.1f |" % (short(k), len(v), sum(v) / 1e3, sum(v) / len(v),
                                                                   min(v), max(v), 100 * sum(v) / tot))
    # ---- PMC passes
    for label, sub, ctr in (("FETCH_SIZE", "pmc_fetch", "FETCH_SIZE"), ("WRITE_SIZE", "pmc_write", "WRITE_SIZE")):
        acc = defaultdict(list)
        for f in find(os.path.join(root, sub), "*counter_collection.csv"):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if r.get("Counter_Name") == ctr:
                        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        if not acc:
            continue
        print("\n## PMC %s (own pass; KiB per dispatch)\n" % label)
        print("| kernel | dispatches | avg KiB | avg bytes | avg bytes x2 (gfx950 read correction) |")
        print("|---|---:|---:|---:|---:|")
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            avg = sum(v) / len(v)
            print("| `%s` | %d | %.1f | %.0f | %s |" % (short(k), len(v), avg, avg * 1024,
                                                     ("%.0f" % (avg * 2048)) if ctr == "FETCH_SIZE" else "-"))
    # ---- SQ / GRBM passes (tools/profile_gpu.sh: pmc_sq_a, pmc_sq_b, pmc_grbm): per-dispatch averages of every counter
    for sub in ("pmc_sq_a", "pmc_sq_b", "pmc_grbm"):
        acc = defaultdict(lambda: defaultdict(list))
        for f in find(os.path.join(root, sub), "*counter_collection.csv"):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if not acc:
            continue
        names = sorted({c for k in acc for c in acc[k]})
        print("\n## PMC %s (own pass; average per dispatch, summed over the chip)\n" % sub)
        print("| kernel | dispatches | " + " | ".join(names) + " |")
        print("|---|---:|" + "---:|" * len(names))
        for k, cs in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
            n = max(len(v) for v in cs.values())
            print("| `%s` | %d | " % (short(k), n) + " | ".join(("%.4g" % (sum(cs[c]) / len(cs[c]))) if c in cs else "-" for c in names) + " |")


if __name__ == "__main__":
    main()
