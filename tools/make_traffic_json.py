#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of tools/profile_round.sh: per-launch HBM bytes ((FETCH_SIZE * 2 +
WRITE_SIZE) * 1024: MI355X_MICROARCH.md, HBM section; calibrated by tools/calib_fetch.hip) and per-launch SQ / TCP
counters of the 10,000-pair match kernels and the configs[2] residual kernel.  usage: make_traffic_json.py <dir>"""
import collections, csv, glob, hashlib, json, os, re, sys

d = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash(root=ROOT):
    """sha256 over the kernel sources the counters were taken from (bench.py recomputes it on the tree it runs in and
    refuses to price its roofline with counters of another build)."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "nautilus_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(root, "nautilus_amd", "csrc", "*.h")) +
                    [os.path.join(root, "nautilus_amd", "csrc", "Makefile")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]



def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n)
    return n.replace("void ", "").replace("nhip::", "")


per = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> per-dispatch values
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))


def med(kernel, counter):
    v = sorted(per.get(kernel, {}).get(counter, []))
    # the full-size launches are the large ones: median of the upper half drops warm-ups of other sizes
    v = [x for x in v if x >= 0.5 * v[-1]] if v else v
    return v[len(v) // 2] if v else None


out = {"source": "tools/profile_round.sh: rocprofv3 --pmc passes of `bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-drop-in`; "
                 "bytes = (FETCH_SIZE*2 + WRITE_SIZE)*1024 per dispatch",
       "kernel_source_hash": kernel_source_hash(),
       "workload": {"mode": "weak", "pairs": 10000, "scans": 1000, "per_target": 10}}
# the branch-and-bound matcher: fused form (one kernel per 10,000-pair launch; NHIP_BNB_SPLIT=0) or split form (bounds +
# seeds, then the candidates): whichever the profiled run launched
names = {"csm_bnb_kernel<1, true, true, false>": ("bnb", 8), "csm_bnb_kernel<2, true, true, false>": ("bnb", 16),
         "csm_bnb_kernel<1, true, true, true>": ("bnb_bounds", 8), "csm_bnb_kernel<2, true, true, true>": ("bnb_bounds", 16),
         "csm_bnb_cand_kernel<1>": ("bnb_cand", 8), "csm_bnb_cand_kernel<2>": ("bnb_cand", 16),
         "csm_correlate_kernel<false, false>": ("correlate", 8), "csm_correlate16_kernel<false, false>": ("correlate", 16)}
for k, (tag, bits) in names.items():
    f, w = med(k, "FETCH_SIZE"), med(k, "WRITE_SIZE")
    if f is not None and w is not None:
        out["csm_%s_bytes_per_launch_10000pairs_u%d" % (tag, bits)] = (2 * f + w) * 1024
        out["csm_%s_write_bytes_per_launch_10000pairs_u%d" % (tag, bits)] = w * 1024
    sq = {c: med(k, c) for c in per.get(k, {}) if c not in ("FETCH_SIZE", "WRITE_SIZE")}
    if sq:
        sq["source"] = "rocprofv3 SQ / TCP counters of %s, per 10,000-pair launch" % k
        out["csm_%s_sq_per_launch_10000pairs_u%d" % (tag, bits)] = sq
k = "resid_lidar_kernel<0, true>"
f, w = med(k, "FETCH_SIZE"), med(k, "WRITE_SIZE")
if f is not None and w is not None:
    out["resid_lidar_bytes_per_launch_10750545corr"] = (2 * f + w) * 1024
print(json.dumps(out, indent=1))
