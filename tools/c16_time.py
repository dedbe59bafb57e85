"""Times the kernel that performs every add on 16-bit cells (csm_correlate16_kernel) for every library under
build/variants/ (tools/c16_variants.sh), one subprocess each (NHIP_LIB), interleaved over `rounds`.
  python tools/c16_time.py [--scans 300] [--rounds 3]        -> JSON lines on stdout"""
import argparse
import glob
import json
import os
os.environ.setdefault("NHIP_TUNABLES", "1")  # (the library reads its switches only then)
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(scans, steps):
    sys.path.insert(0, ROOT)
    import ctypes as C
    import torch
    import bench
    from nautilus_amd import _lib, sharding
    lib = _lib.load()
    wl = bench.Workload("weak", 1, scans=scans)
    plan = sharding.ShardPlan(wl.src, wl.tgt, wl.th0, 1)
    dev = torch.device("cuda", 0)
    out = {}
    for dense in ("0", "1"):
        os.environ["NHIP_CSM_DENSE"] = dense
        m = bench.HipMatcher(wl, plan.shard(0), dev, 16, exhaustive=True)
        m.step()
        torch.cuda.synchronize()
        lib.nhip_timing_reset()
        lib.nhip_timing_enable(1)
        for _ in range(steps):
            m.step()
        torch.cuda.synchronize()
        lib.nhip_timing_enable(0)
        ms, n = C.c_double(0), C.c_int32(0)
        lib.nhip_timing_get(0, C.byref(ms), C.byref(n))
        out["dense" if dense == "1" else "skip"] = ms.value / max(n.value, 1)
        out["sum_check"] = int(m.d_sums[:m.n_pairs].sum().item())
        m.free_grids()
    out["pairs"] = wl.n_pairs
    print(json.dumps(out))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=300)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        child(a.scans, a.steps)
        sys.exit(0)
    libs = sorted(glob.glob(os.path.join(ROOT, "build", "variants", "lib_*.so")))
    for r in range(a.rounds):
        for lp in libs:
            env = dict(os.environ, NHIP_LIB=lp)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--scans", str(a.scans), "--steps", str(a.steps)],
                               env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            line = p.stdout.decode().strip().splitlines()[-1] if p.stdout.strip() else p.stderr.decode()[-400:]
            print(json.dumps({"lib": os.path.basename(lp), "round": r, "result": line}), flush=True)
