"""Where corr_search_kernel's time goes: the product build and builds with parts of the bucket walk compiled out
(-DNHIP_CORR_EXPERIMENT=1: no walk; =2: the walk without its point reads; results WRONG), one subprocess each (GPU box)."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, ROOT)
    import bench
    from nautilus_amd import synth, csm
    bag = synth.SynthBag(1000, dense=True)
    xy, off = csm.pack_scans(bag.scans)
    r = bench.bench_icp(bag, xy, off, False)
    print("corr_search_ms %.4f normal_eq_ms %.4f" % (r["corr_search_ms"], r["normal_eq_ms"]))
    sys.exit(0)
for lib in [None] + sorted(glob.glob(os.path.join(ROOT, "build", "variants", "libcorr_*.so"))):
    env = dict(os.environ)
    if lib:
        env["NHIP_LIB"] = lib
    p = subprocess.run([sys.executable, __file__, "one"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    print(os.path.basename(lib) if lib else "product", p.stdout.decode().strip().splitlines()[-1:], flush=True)
