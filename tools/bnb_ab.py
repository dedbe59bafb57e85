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
                res.setdefault(name, {}).setdefault(line.split()[0], []).append(float(line.split()[2]))
        print(name, r, res.get(name), flush=True)
print(json.dumps(res))
