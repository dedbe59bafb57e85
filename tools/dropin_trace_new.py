"""GetTransformation on targets the cache has not seen (for a rocprofv3 --kernel-trace run: where a new target's call goes)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nautilus_amd import csm, synth
bag = synth.SynthBag(60, dense=True)
m = csm.CorrelativeScanMatcher(30, 2, 0.3, 0.01)
for t in (10, 20, 30, 40, 50):
    m.GetTransformation(bag.scans[t + 2], bag.scans[t], bag.odom[t + 2, 2], bag.odom[t, 2], math.radians(90))
