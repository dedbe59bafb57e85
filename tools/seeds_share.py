"""Share of the bounds kernel's workgroup time spent in the seeds (instrumented build, the bench's 10,000 pairs)."""
import os, sys
os.environ["NHIP_TUNABLES"] = "1"; os.environ["NHIP_BNB_INSTRUMENT"] = "1"; os.environ["NHIP_BNB_STATS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from nautilus_amd import csm, sharding
wl = bench.Workload("weak", 1)
plan = sharding.ShardPlan(wl.src, wl.tgt, wl.th0, 1, None)
m = bench.HipMatcher(wl, plan.shard(0), torch.device("cuda:0"), 16)
m.step(); torch.cuda.synchronize(); csm.bnb_stats_levels()
m.step(); torch.cuda.synchronize()
lv = csm.bnb_stats_levels()
print({k: lv[k] for k in ("clk_bounds", "clk_seeds", "blocks_whole", "sub_blocks", "candidates_refined")})
print("seeds / (bounds + seeds) of the first kernel's workgroup time: %.3f" % (lv["clk_seeds"] / float(lv["clk_seeds"] + lv["clk_bounds"])))
