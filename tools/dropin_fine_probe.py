"""Cached GetTransformation calls, fine level by the branch-and-bound matcher (NHIP_DROPIN_FINE=bnb) vs by every add
(NHIP_DROPIN_FINE=every_add = the library's default form: the kernel whose lanes are poses in tiles of rows; profiles/
r06_dropin_fine_level.txt was taken when that value still meant the STRIP kernels, now NHIP_DROPIN_FINE=strips), against the coarse
optimum's score (GPU box)."""
import os, sys, time, math, ctypes as C
os.environ["NHIP_TUNABLES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nautilus_amd import csm, synth, _lib
bag = synth.SynthBag(160, dense=True)
m = csm.CorrelativeScanMatcher(30, 2, 0.3, 0.01)
lib = _lib.load()
info = (C.c_double * 4)()
rows = []
for tgt in (40, 100):
    for src in list(range(tgt + 1, tgt + 14)) + list(range(tgt + 14, tgt + 50, 3)):
        a = (bag.scans[src], bag.scans[tgt], bag.odom[src, 2], bag.odom[tgt, 2], math.radians(90))
        t = {}
        for mode in ("bnb", "every_add"):
            os.environ["NHIP_DROPIN_FINE"] = mode
            r = m.GetTransformation(*a)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); r = m.GetTransformation(*a); ts.append(time.perf_counter() - t0)
            t[mode] = 1e3 * float(np.median(ts))
            lib.nhip_csm_get_transformation_info(info)
        d = float(np.hypot(*(bag.truth[src, :2] - bag.truth[tgt, :2]))) if hasattr(bag, "truth") else -1.0
        rows.append((info[0], t["bnb"], t["every_add"], float(r[0]), tgt, src, d))
        print("tgt %3d src %3d dist %5.2f coarse %8.3f final %8.3f  bnb %.3f ms  every_add %.3f ms" % (tgt, src, d, info[0], float(r[0]), t["bnb"], t["every_add"]), flush=True)
rows.sort()
print("\nsorted by coarse score:")
for r in rows:
    print("coarse %8.3f  bnb %.3f  every_add %.3f  %s" % (r[0], r[1], r[2], "<- every_add faster" if r[2] < r[1] else ""))
for thr in (-0.5, -0.75, -1.0, -1.25, -1.5, -2.0, -2.5, -3.0, -4.0, -1e9):
    tot = sum((r[2] if r[0] < thr else r[1]) for r in rows)
    print("threshold %7.2f: mean ms per call %.3f" % (thr, tot / len(rows)))
print("always bnb %.3f, always every_add %.3f, oracle choice %.3f" % (sum(r[1] for r in rows) / len(rows), sum(r[2] for r in rows) / len(rows), sum(min(r[1], r[2]) for r in rows) / len(rows)))
