"""Where the fine level's branch-and-bound kernel spends a cached GetTransformation call (instrumented build's shader-clock sums)."""
import os, sys, math
os.environ["NHIP_TUNABLES"] = "1"; os.environ["NHIP_BNB_INSTRUMENT"] = "1"; os.environ["NHIP_BNB_STATS"] = "1"; os.environ["NHIP_DROPIN_FINE"] = "bnb"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nautilus_amd import csm, synth
bag = synth.SynthBag(120, dense=True)
m = csm.CorrelativeScanMatcher(30, 2, 0.3, 0.01)
for src in (42, 48, 52):
    a = (bag.scans[src], bag.scans[40], bag.odom[src, 2], bag.odom[40, 2], math.radians(90))
    m.GetTransformation(*a)
    csm.bnb_stats_levels()
    N = 10
    for _ in range(N):
        m.GetTransformation(*a)
    lv = csm.bnb_stats_levels()
    # (coarse level runs the every-add small-plane kernel: the counters are the fine level's three workgroups)
    print("source", src, {k: round(v / N / (3 if k in ("clk_bounds", "clk_seeds", "clk_slowest_wave") else 1), 1) for k, v in lv.items()}, flush=True)
print("clk_bounds / clk_seeds / clk_slowest_wave: per workgroup (three per call), shader clocks; clk_wave_phase3 etc.: sums over waves; 100 MHz ticks where named")
