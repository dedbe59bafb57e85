"""Cached GetTransformation calls with the fine level dealt over 1 / 3 / 7 workgroups and fewer rotations kept by a pair's
own workgroup (GPU box)."""
import os, sys, time, math, subprocess
os.environ.setdefault("NHIP_TUNABLES", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import numpy as np
    from nautilus_amd import csm, synth
    bag = synth.SynthBag(120, dense=True)
    m = csm.CorrelativeScanMatcher(30, 2, 0.3, 0.01)
    args = lambda i, j: (bag.scans[i], bag.scans[j], bag.odom[i, 2], bag.odom[j, 2], math.radians(90))
    out = []
    for i in (42, 48, 52, 60):
        m.GetTransformation(*args(i, 40))
        ts = []
        for r in range(8):
            t0 = time.perf_counter(); m.GetTransformation(*args(i, 40)); ts.append(time.perf_counter() - t0)
        out.append("%.3f" % (1e3 * float(np.median(ts))))
    print(" ".join(out))
    sys.exit(0)
for env in ({"NHIP_DROPIN_PARTS": "1"}, {}, {"NHIP_DROPIN_PARTS": "7"}, {"NHIP_BNB_KEEP_RANKS": "2"}, {"NHIP_BNB_KEEP_RANKS": "4"},
            {"NHIP_DROPIN_PARTS": "7", "NHIP_BNB_KEEP_RANKS": "1"}, {"NHIP_BNB_KEEP_RANKS": "0"}):
    p = subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    print(env, "ms per call, sources 42 48 52 60 of target 40:", p.stdout.decode().strip().splitlines()[-1:], flush=True)
