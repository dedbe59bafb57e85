import os, sys, time, math
os.environ.setdefault("NHIP_TUNABLES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nautilus_amd import csm, synth, _lib
bag = synth.SynthBag(120, dense=True)
m = csm.CorrelativeScanMatcher(30, 2, 0.3, 0.01)
args = lambda i, j: (bag.scans[i], bag.scans[j], bag.odom[i, 2], bag.odom[j, 2], math.radians(90))
lib = _lib.load()
m.GetTransformation(*args(42, 40))
for i in (41, 42, 44, 48, 52, 60):
    ts = []
    for r in range(5):
        t0 = time.perf_counter(); m.GetTransformation(*args(i, 40)); ts.append(time.perf_counter() - t0)
    print("source", i, "target 40: ms per cached call", ["%.3f" % (1e3 * t) for t in ts], csm.drop_in_cache_stats())
# kernel time inside
lib.nhip_timing_reset(); lib.nhip_timing_enable(1)
for r in range(10): m.GetTransformation(*args(42, 40))
lib.nhip_timing_enable(0)
import ctypes as C
ms, n = C.c_double(0), C.c_int32(0)
lib.nhip_timing_get(_lib.NHIP_TIMER_CSM, C.byref(ms), C.byref(n))
print("matcher kernels: %.3f ms per launch over %d launches" % (ms.value / max(n.value, 1), n.value))
for mode in ("bnb", "every_add"):
    os.environ["NHIP_TUNABLES"] = "1"
    if mode == "bnb":
        os.environ["NHIP_DROPIN_COARSE"] = "bnb"
    else:
        os.environ.pop("NHIP_DROPIN_COARSE", None)
    ts = []
    for r in range(12):
        t0 = time.perf_counter(); m.GetTransformation(*args(42 + r % 4, 40)); ts.append(time.perf_counter() - t0)
    print("coarse level by", mode, ": median ms per cached call %.3f" % (1e3 * float(np.median(ts))))
