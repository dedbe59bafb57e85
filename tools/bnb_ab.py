"""A/B of branch-and-bound matcher builds: tools/bnb_quick.py (kernel ms on the bench workload, both cell widths) for
every build/variants/libbnb_*.so (tools/bnb_variants.sh), one subprocess each, interleaved over the rounds.
  python tools/bnb_ab.py [rounds]"""
import glob, json, os, subprocess, sys
os.environ.setdefault("NHIP_TUNABLES", "1")  # (the library reads its switches only then)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
libs = sorted(glob.glob(os.path.join(ROOT, "build", "variants", "libbnb_*.so")))
res = {}
for r in range(rounds):
    for lp in libs:
      for order in ("0",):  # ("0", "1"): by-target launch order / heaviest first
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bnb_quick.py")],
                           env=dict(os.environ, NHIP_LIB=lp, NHIP_QUICK_ORDER=order), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        name = os.path.basename(lp)[6:-3] + ("+lpt" if order == "1" else "")
        for line in p.stdout.decode().splitlines():
            if "kernel_ms" in line:
                f = line.split()
                res.setdefault(name, {}).setdefault(f[0], []).append(float(f[2]))
                if len(f) >= 9:  # bounds / candidates kernels of the split form, checksum of the records
                    res[name].setdefault(f[0] + "_bounds", []).append(float(f[4]))
                    res[name].setdefault(f[0] + "_cand", []).append(float(f[6]))
                    res[name].setdefault(f[0] + "_crc", set()).add(f[8])
        print(name, r, res.get(name), flush=True)
print(json.dumps({k: {kk: (sorted(vv) if isinstance(vv, set) else vv) for kk, vv in v.items()} for k, v in res.items()}))
