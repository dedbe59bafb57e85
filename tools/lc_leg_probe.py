"""Which 200-scan bag makes bench.py's configs[0] leg close its loop?  (GPU box; prints one line per spacing)"""
import json, os, sys
os.environ.setdefault("NHIP_TUNABLES", "1")  # (the library reads its switches only then)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import slam_loop
slam_loop.run(n_scans=40, window=2, hitl=False, min_scatter_score=0.3)
for spacing in (0.3, 0.35, 0.4, 0.45, 0.5, 0.55):
    for seed in (20201114, 7):
        r = slam_loop.run(n_scans=200, window=10, min_scatter_score=0.3, cell_bits=16, spacing=spacing, seed=seed)
        print(json.dumps({"spacing": spacing, "seed": seed, **{k: r.get(k) for k in ("lc_candidate_scans", "lc_candidates", "lc_accepted", "lc_rel_err_m", "err_icp_m", "err_lc_m", "err_hitl_m", "t_csm_s", "t_total_s")}}), flush=True)
