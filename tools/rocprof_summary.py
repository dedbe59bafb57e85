#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats / PMC counters) with short kernel names.
usage: rocprof_summary.py <dir> [--pmc]"""
import csv, glob, os, re, sys, collections

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n)
    n = n.replace("void ", "").replace("nhip::", "")
    return n[:60]

d = sys.argv[1]
if "--per-dispatch" in sys.argv:
    # one line per dispatch of kernels whose name contains the given substring: counters side by side
    sub = sys.argv[sys.argv.index("--per-dispatch") + 1]
    rows = collections.OrderedDict()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub not in r["Kernel_Name"]:
                continue
            key = (int(r["Dispatch_Id"]), short(r["Kernel_Name"]))
            rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    names = sorted({c for v in rows.values() for c in v})
    print("dispatch kernel " + " ".join(names))
    for (did, k), v in sorted(rows.items()):
        print("%d %s " % (did, k) + " ".join("%.4g" % v.get(c, float("nan")) for c in names))
elif "--pmc" in sys.argv:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    for k in agg:
        if "at::" in k or "rocclr" in k:
            continue
        print(k)
        for c, v in sorted(agg[k].items()):
            n = cnt[(k, c)]
            print("   %-28s total %.6g  per-dispatch %.6g  (n=%d)" % (c, v, v / n, n))
else:
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        print("# %s" % f)
        print("%-62s %6s %14s %14s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
        for r in csv.DictReader(open(f)):
            print("%-62s %6s %14s %14.0f %7s" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"],
                                                 float(r["AverageNs"]), r["Percentage"]))
