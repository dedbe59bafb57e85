"""One target, 40 cached GetTransformation calls (for a rocprofv3 --kernel-trace --stats run: per-kernel time of a call)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nautilus_amd import csm, synth
bag = synth.SynthBag(60, dense=True)
m = csm.CorrelativeScanMatcher(30, 2, 0.3, 0.01)
for r in range(41):
    i = 42 + r % 4
    m.GetTransformation(bag.scans[i], bag.scans[40], bag.odom[i, 2], bag.odom[40, 2], math.radians(90))
