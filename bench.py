#!/usr/bin/env python3
"""bench.py -- loop-closure candidate pairs/s on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input on every rank:
  host: (cos, sin) of each pair's odometry heading difference (16 B/pair) + async H2D
  K1  : likelihood grids of the batch's target scans          (nhip_grid_build_dev)
  K2/3: exhaustive (theta, x, y) correlation + argmax per pair (nhip_csm_match_dev)
  N>1 : ONE RCCL all-gather of the 16-byte best-pose records   (torch.distributed, backend nccl)
Workload at every N (weak scaling): BASELINE configs[1] per GPU -- 1,000 dense 1081-beam scans,
10,000 candidate pairs (10 per target), 61 x 81 x 81 lattice (1 deg / 5 cm over +-30 deg / +-2 m),
1200 x 1200 grid at 0.05 m.  Scans, pair list and odometry are resident in HBM before the timed
region; the PCIe-inclusive figure is discussed in DESIGN.md.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _traffic(key):
    """HBM bytes per launch from the committed PMC profile (profiles/traffic.json), or None."""
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(tf)).get(key)
    except Exception:
        return None


def _onchip(n_pairs, n_theta, avg_ms):
    """VALU-, SALU- and LDS-issue utilisation of csm_correlate_kernel: the instruction counts rocprofv3's SQ
    counters measured for this workload (profiles/traffic.json <- profiles/r01_pmc_sq_correlate.txt, per
    10k-pair launch, scaled by the pair count) divided by the kernel time measured live.  LDS read bytes:
    every LDS instruction that is not a tile-fill store (3 per 16-byte fill load) moves 7/4 dwords per lane.
    Peaks (MI355X_MICROARCH.md): ds_read_b32 128 B/clk/CU -> ~75 TB/s chip; VALU one wave64 instruction per
    2 clk per SIMD -> 256 CU x 4 SIMD x 2.4 GHz / 2 = 1.23e12 wave-instr/s; SALU one instruction per clk per
    CU -> 6.1e11/s."""
    sq = _traffic("csm_correlate_sq_per_launch_10000pairs")
    if not sq or n_theta != 61:
        return None
    k = n_pairs / 10000.0
    secs = avg_ms * 1e-3
    valu, salu = k * sq["SQ_INSTS_VALU"] / secs, k * sq["SQ_INSTS_SALU"] / secs
    waves = n_pairs * n_theta * 4.0
    fill_loads = k * sq["SQ_INSTS_VMEM_RD"] - waves * 2 * 17  # minus the point and skip-byte loads of 17 lane-chunks
    lds_reads = k * sq["SQ_INSTS_LDS"] - 3.0 * fill_loads
    lds = lds_reads * (7.0 / 4.0) * 256.0 / secs / 1e12
    # SQ_ACTIVE_INST_VALU counts 4-clock issue slots; 1024 SIMDs x kernel time x clock / 4 are available
    active = sq.get("SQ_ACTIVE_INST_VALU")
    valu_busy = (k * active) / (1024.0 * secs * 2.4e9 / 4.0) if active else None
    return {"valu_issue_slots_busy": valu_busy, "lds_read_TBps": lds, "lds_read_peak_TBps": 75.0, "lds_frac": lds / 75.0,
            "valu_wave_instr_per_s": valu, "valu_peak_wave_instr_per_s": 1.2288e12, "valu_frac": valu / 1.2288e12,
            "salu_instr_per_s": salu, "salu_peak_instr_per_s": 6.144e11, "salu_frac": salu / 6.144e11,
            "wave_wait_frac": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scans", type=int, default=1000)
    ap.add_argument("--per-target", type=int, default=10)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-resid", action="store_true", help="skip the secondary residual-kernel measurement")
    ap.add_argument("--no-drop-in", action="store_true",
                    help="skip the single-pair latency leg (its small launches of the correlation kernel would "
                         "blur that kernel's average in a rocprofv3 --stats summary)")
    return ap.parse_args()


def cpu_baseline(bag, xy, off, ids, src, slot, th0, budget_s):
    """The oracle (CPU restatement, OpenMP over pairs like the reference's -fopenmp build) timed on
    this host's cores on a bounded sample of the same workload: whole targets (grid build + their
    pairs), sized from a one-pair calibration to ~budget_s seconds."""
    from oracle import oracle as O
    ospec = O.grid_spec()
    oss = O.search_spec(61, 81, 81, math.radians(1.0))
    cores = O.num_threads()
    t0 = time.perf_counter()
    g0 = O.grid_build_batch(xy, off, ids[:1], ospec, 1)
    O.csm_match_batch(xy, off, g0, ospec, src[:1], np.zeros(1, np.int32), th0[:1], oss, None, 1)
    t_one = time.perf_counter() - t0  # one grid + one pair on one core
    per_target = int(np.sum(slot == 0))
    t_target = t_one * (1 + per_target) / 2.0  # grid ~ pair cost, amortised below by measuring
    n_targets = int(max(1, min(len(ids), (budget_s * cores) / max(t_target, 1e-3))))
    n_targets = max(cores // max(per_target, 1), n_targets)
    n_targets = min(n_targets, len(ids), 400)  # oracle grids are 1.44 MB each, keep host memory small
    sel = np.nonzero(slot < n_targets)[0]
    t0 = time.perf_counter()
    grids = O.grid_build_batch(xy, off, ids[:n_targets], ospec, cores)
    t_grid = time.perf_counter() - t0
    t0 = time.perf_counter()
    res = O.csm_match_batch(xy, off, grids, ospec, src[sel], slot[sel], th0[sel], oss, None, cores)
    t_match = time.perf_counter() - t0
    return {"value": len(sel) / (t_grid + t_match), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "%d pairs / %d targets of the same workload (grid build %.2f s + match %.2f s), "
                      "oracle C restatement, OpenMP over pairs" % (len(sel), n_targets, t_grid, t_match),
            "single_thread_pairs_per_s": 1.0 / max(t_one, 1e-9),
            "build_flags": "-O3 -fopenmp -DNDEBUG (the reference's CMakeLists.txt:16), -ffp-contract=off",
            "cpu_model": _cpu_model(),
            }, sel, res


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    from nautilus_amd import _lib, csm, synth
    lib = _lib.load()
    assert torch.cuda.is_available(), "bench.py needs an MI355X: there is no CPU path"
    torch.cuda.set_device(local)
    _lib.check(lib.nhip_set_device(local))
    dev = torch.device("cuda", local)
    use_dist = "RANK" in os.environ  # launched by torch.distributed.run: RCCL group even at N=1
    if use_dist:
        # RCCL prints a version banner on stdout when the communicator is created; keep stdout to
        # the one JSON line by routing fd 1 to stderr until the first collective has run.
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    # ---- synthetic workload (per rank; seeds differ per rank so shards are not copies)
    bag = synth.SynthBag(a.scans, dense=True, seed=synth.SEED + 1000 * rank)
    assert all(len(s) == synth.N_BEAMS for s in bag.scans), "dense world must return all 1081 beams"
    xy, off = csm.pack_scans(bag.scans)
    ids = np.arange(a.scans, dtype=np.int32)
    src, tgt, th0 = bag.sample_pairs(per_target=a.per_target, targets=ids, max_dist=1.5, min_sep=20,
                                     seed=synth.SEED + 1000 * rank)
    slot = tgt.astype(np.int32)  # target i -> grid slot i; pairs are already sorted by target
    n_pairs = len(src)
    spec = csm.grid_spec(30.0, 0.05, 2.0, 1e-10, 40)
    search = csm.search_spec(61, 81, 81, math.radians(1.0))
    L = csm.grid_layout(spec)

    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d_xy, d_off, d_ids, d_src, d_slot = t(xy), t(off), t(ids), t(src), t(slot)
    d_delta = t(csm.delta_table(search))
    h_th0 = np.ascontiguousarray(th0, dtype=np.float64)
    h_rot0 = torch.empty((n_pairs, 2), dtype=torch.float64).pin_memory()
    d_rot0 = torch.empty((n_pairs, 2), dtype=torch.float64, device=dev)
    d_grids = torch.empty(lib.nhip_grids_bytes(C.byref(spec), len(ids)), dtype=torch.uint8, device=dev)
    chunk = len(ids)  # the grid-build workspace is a few KB per target: all targets in one pass
    ws_bytes = lib.nhip_grid_workspace_bytes(C.byref(spec), chunk)
    d_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    d_keys = torch.empty(n_pairs, dtype=torch.int64, device=dev)
    d_out = torch.empty((n_pairs, 4), dtype=torch.int32, device=dev)
    d_sums = torch.empty(n_pairs, dtype=torch.int32, device=dev)
    d_all = torch.empty((world * n_pairs, 4), dtype=torch.int32, device=dev) if use_dist else None
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    rot0_np = h_rot0.numpy()

    def step():
        _lib.check(lib.nhip_csm_rot0(_lib.ptr(h_th0), None, n_pairs, _lib.ptr(rot0_np)))
        d_rot0.copy_(h_rot0, non_blocking=True)
        _lib.check(lib.nhip_grid_build_dev(d_xy.data_ptr(), d_off.data_ptr(), d_ids.data_ptr(), len(ids),
                                           C.byref(spec), d_grids.data_ptr(), d_ws.data_ptr(), ws_bytes, sp))
        _lib.check(lib.nhip_csm_match_dev(d_xy.data_ptr(), d_off.data_ptr(), d_grids.data_ptr(), C.byref(spec),
                                          d_src.data_ptr(), d_slot.data_ptr(), d_rot0.data_ptr(),
                                          d_delta.data_ptr(), None, n_pairs, C.byref(search),
                                          d_keys.data_ptr(), d_out.data_ptr(), d_sums.data_ptr(), sp))
        if use_dist:
            dist.all_gather_into_tensor(d_all, d_out)  # the ONE collective: 16 B per pair

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    lib.nhip_timing_reset()
    lib.nhip_timing_enable(1)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    lib.nhip_timing_enable(0)
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    k_ms, k_n = C.c_double(0), C.c_int32(0)
    _lib.check(lib.nhip_timing_get(_lib.NHIP_TIMER_CSM, C.byref(k_ms), C.byref(k_n)))
    g_ms, g_n = C.c_double(0), C.c_int32(0)
    _lib.check(lib.nhip_timing_get(_lib.NHIP_TIMER_GRID, C.byref(g_ms), C.byref(g_n)))

    if use_dist:
        # every rank's block of the gathered table must equal what that rank computed
        mine = d_all[rank * n_pairs:(rank + 1) * n_pairs]
        assert torch.equal(mine, d_out), "all-gather returned a different block for this rank"
    if rank != 0:
        dist.destroy_process_group()
        return

    # ---- parity spot check of the timed result against the oracle happens inside cpu_baseline
    got = d_out.cpu().numpy().view(csm.MATCH_DTYPE).reshape(-1)
    got_sums = d_sums.cpu().numpy()
    lookups_per_pair = search.n_theta * search.nx * search.ny * synth.N_BEAMS  # 432,638,901
    bytes_per_launch = float(n_pairs) * lookups_per_pair * 1  # 1-byte cells
    avg_ms = k_ms.value / max(k_n.value, 1)
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    traffic = _traffic("csm_correlate_bytes_per_launch_%dpairs" % n_pairs)
    out = {
        "metric": "loop-closure candidate pairs/sec (1081-beam)",
        "value": world * n_pairs * a.steps / elapsed,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / a.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1] per GPU: %d dense 1081-beam scans, %d candidate pairs "
                               "(%d per target), 61x81x81 lattice (1 deg / 5 cm over +-30 deg / +-2 m), "
                               "1200x1200 u8 log-likelihood grid at 0.05 m; grid build + match + all-gather"
                               % (a.scans, n_pairs, a.per_target),
                   "pairs_per_gpu": n_pairs, "scans_per_gpu": a.scans, "lattice": [61, 81, 81],
                   "grid": [L.side, L.side], "cell_bytes": 1, "collective": "all_gather 16 B/pair" if world > 1 else "none"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "csm_correlate_kernel<false, false>", "avg_launch_ms": avg_ms, "launches": k_n.value,
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "note": "algorithmic gather bytes (1 B per grid lookup) / kernel time; the grid is "
                             "cache/LDS-resident, so this exceeds what HBM itself moves (see traffic)"},
        "kernels_ms_per_step": {"csm_correlate": k_ms.value / a.steps, "grid_blur_and_skipmap": g_ms.value / a.steps},
        # what actually bounds the correlate kernel: the tile is LDS-resident, so
        # the honest ceilings are the LDS read pipe and VALU issue, not HBM.
        "onchip_roofline": _onchip(n_pairs, search.n_theta, avg_ms),
    }
    if world == 1:
        # the same steps with the skip map ignored (every add of the exhaustive definition performed)
        try:
            os.environ["NHIP_CSM_DENSE"] = "1"
            step()
            fence()
            lib.nhip_timing_reset()
            lib.nhip_timing_enable(1)
            t0 = time.perf_counter()
            for _ in range(2):
                step()
            fence()
            dt = (time.perf_counter() - t0) / 2
            lib.nhip_timing_enable(0)
            dk_ms, dk_n = C.c_double(0), C.c_int32(0)
            _lib.check(lib.nhip_timing_get(_lib.NHIP_TIMER_CSM, C.byref(dk_ms), C.byref(dk_n)))
            dense_kernel_ms = dk_ms.value / max(dk_n.value, 1)
            dense_gbs = bytes_per_launch / (dense_kernel_ms * 1e-3) / 1e9
            same = bool(np.array_equal(d_sums.cpu().numpy(), got_sums) and
                        d_out.cpu().numpy().tobytes() == got.tobytes())
            out["zero_skip"] = {"dense_value": n_pairs / dt, "dense_ms_per_step": 1e3 * dt, "same_result": same,
                                "dense_kernel_ms": dense_kernel_ms, "dense_roofline_achieved_GBps": dense_gbs,
                                "dense_roofline_frac": dense_gbs / HBM_PEAK_GBS,
                                "note": "NHIP_CSM_DENSE=1: all-zero window strips are added like any other; `value` "
                                        "leaves them out (skip map built with the grids), results are identical"}
        finally:
            os.environ.pop("NHIP_CSM_DENSE", None)
    if world == 1 and a.cpu_seconds > 0:
        cb, sel, ref = cpu_baseline(bag, xy, off, ids, src, slot, h_th0, a.cpu_seconds)
        ok = all(np.array_equal(got[f][sel], ref[f]) for f in ("itheta", "ix", "iy")) and \
            np.array_equal(got_sums[sel], ref["sum"])
        cb["gpu_matches_oracle_on_sample"] = bool(ok)
        out["cpu_baseline"] = cb
        if not ok:
            print("PARITY FAILURE: GPU result differs from the oracle on the cpu_baseline sample", file=sys.stderr)
    if world == 1 and not a.no_resid:
        try:
            out["secondary"] = {"resid_lidar": bench_residuals(torch, lib, dev, sp, a.cpu_seconds > 0)}
        except Exception as e:  # secondary measurement must not lose the headline line
            out["secondary"] = {"resid_lidar_error": repr(e)}
        try:
            del d_grids
            torch.cuda.empty_cache()
            out["secondary"]["icp_front_half"] = bench_icp(bag, xy, off, a.cpu_seconds > 0)
        except Exception as e:
            out["secondary"]["icp_front_half_error"] = repr(e)
        try:
            # PCIe-inclusive: the handle API (host buffers in, host records out; it also allocates and
            # frees its device memory per call) on the same workload -- never the headline `value`.
            t0 = time.perf_counter()
            st = csm.ScanTable(xy, off)
            gr = csm.LikelihoodGrids(st, ids, spec)
            hm, hs = csm.match_pairs(st, gr, src, slot, h_th0, search)
            dt = time.perf_counter() - t0
            gr.close()
            st.close()
            out["secondary"]["host_buffer_api"] = {
                "pairs_per_s": n_pairs / dt, "seconds": dt,
                "same_result_as_device_api": bool(np.array_equal(hs, got_sums) and hm.tobytes() == got.tobytes()),
                "note": "nhip_scans_upload + nhip_grids_build + nhip_csm_match with host pointers, incl. "
                        "hipMalloc/hipFree and PCIe copies (8.6 MB in, 0.2 MB out)"}
        except Exception as e:
            out["secondary"]["host_buffer_api_error"] = repr(e)
    if world == 1 and not a.no_resid and not a.no_drop_in:
        try:
            out["secondary"]["drop_in_two_level"] = bench_drop_in(bag, a.cpu_seconds > 0)
        except Exception as e:
            out["secondary"]["drop_in_two_level_error"] = repr(e)
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


def bench_drop_in(bag, with_cpu, calls=8):
    """The one-pair call of the reference, CorrelativeScanMatcher(30, 2, 0.3, 0.01).GetTransformation(...)
    (solver.cc:633-638): coarse search on a 0.3 m grid, refinement on the 0.01 m (6000 x 6000) grid around the
    coarse optimum -- latency per call through the host-buffer API, next to the same two searches on the CPU
    oracle (the "csm-style restatement" of SURVEY 8d; the real third_party/csm is not in the tree)."""
    from nautilus_amd import csm
    m = csm.CorrelativeScanMatcher(30, 2, 0.3, 0.01)
    pairs = [(40 + 7 * i, 37 + 7 * i) for i in range(calls)]
    m.GetTransformation(bag.scans[pairs[0][0]], bag.scans[pairs[0][1]], bag.odom[pairs[0][0], 2],
                        bag.odom[pairs[0][1], 2], math.radians(90))
    t0 = time.perf_counter()
    res = [m.GetTransformation(bag.scans[i], bag.scans[j], bag.odom[i, 2], bag.odom[j, 2], math.radians(90))
           for i, j in pairs]
    dt = (time.perf_counter() - t0) / calls
    out = {"workload": "%d single-pair calls on dense 1081-beam scans: 181x13x13 lattice on a 200x200 grid, then "
                       "21x61x61 on a 6000x6000 grid (36 MB built per call)" % calls,
           "seconds_per_call": dt, "calls_per_s": 1.0 / dt}
    if with_cpu:
        from oracle import oracle as O
        i, j = pairs[0]
        a_, b_ = bag.scans[i], bag.scans[j]
        t0 = time.perf_counter()
        theta0 = float(csm.angle_mod(np.float64(bag.odom[i, 2]) - np.float64(bag.odom[j, 2])))
        g1s = O.grid_spec(30.0, 0.3, 2.0, 1e-10)
        m1 = O.csm_match(a_, O.grid_build(b_, g1s), g1s, theta0, O.search_spec(181, 13, 13, math.radians(1.0)))
        tx1, ty1 = np.float32((m1.ix - 6) * 0.3), np.float32((m1.iy - 6) * 0.3)
        th1 = np.float32(theta0 + (m1.itheta - 90) * math.radians(1.0))
        cx, cy = int(round(float(tx1) / 0.01)), int(round(float(ty1) / 0.01))
        g2s = O.grid_spec(30.0, 0.01, 2.0, 1e-10)
        m2 = O.csm_match(a_, O.grid_build(b_, g2s), g2s, float(th1), O.search_spec(21, 61, 61, math.radians(0.1)),
                         (cx, cy))
        dc = time.perf_counter() - t0
        same = bool(np.float32(res[0][0]) == np.float32(m2.score) and
                    res[0][1][0][0] == np.float32((cx + m2.ix - 30) * 0.01) and
                    res[0][1][0][1] == np.float32((cy + m2.iy - 30) * 0.01))
        out["cpu_baseline"] = {"value": 1.0 / dc, "unit": "calls/s", "cores": 1, "kind": "port",
                               "sample": "1 call, oracle C restatement of the same two-level search, single thread",
                               "gpu_matches_oracle_on_sample": same}
    return out


def bench_residuals(torch, lib, dev, sp, with_cpu=False, n_blocks=9945, n_per=1081, iters=20):
    """BASELINE configs[2] shape: 9,945 (i, j) blocks x 1081 correspondences, residual + both
    Jacobians (144 B per correspondence), synthetic correspondences already in HBM."""
    from nautilus_amd import _lib
    n_corr = n_blocks * n_per
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    corr = torch.randn((n_corr, 8), device=dev, dtype=torch.float32, generator=g)
    cb = torch.arange(n_blocks, device=dev, dtype=torch.int32).repeat_interleave(n_per)
    bs = (torch.arange(n_blocks, device=dev, dtype=torch.int32) % 999) + 1
    bt = bs - 1
    poses = torch.randn((1000, 3), device=dev, dtype=torch.float64, generator=g)
    consts = torch.empty(8 * n_blocks, device=dev, dtype=torch.float64)
    res = torch.empty(2 * n_corr, device=dev, dtype=torch.float64)
    js = torch.empty(6 * n_corr, device=dev, dtype=torch.float64)
    jt = torch.empty(6 * n_corr, device=dev, dtype=torch.float64)

    def run():
        _lib.check(lib.nhip_resid_lidar_dev(0, corr.data_ptr(), cb.data_ptr(), n_corr, bs.data_ptr(), bt.data_ptr(),
                                            n_blocks, poses.data_ptr(), 1000, consts.data_ptr(), res.data_ptr(),
                                            js.data_ptr(), jt.data_ptr(), sp))
    run()
    torch.cuda.synchronize()
    lib.nhip_timing_reset()
    lib.nhip_timing_enable(1)
    for _ in range(iters):
        run()
    torch.cuda.synchronize()
    lib.nhip_timing_enable(0)
    ms, n = C.c_double(0), C.c_int32(0)
    _lib.check(lib.nhip_timing_get(_lib.NHIP_TIMER_RESID, C.byref(ms), C.byref(n)))
    avg = ms.value / max(n.value, 1)
    bytes_alg = 144.0 * n_corr
    gbs = bytes_alg / (avg * 1e-3) / 1e9
    out = {"workload": "configs[2]: %d blocks x %d correspondences, LIDARNormal residual + 2 Jacobians" % (n_blocks, n_per),
           "correspondences_per_s": n_corr / (avg * 1e-3), "avg_launch_ms": avg,
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": gbs / HBM_PEAK_GBS, "traffic": _traffic("resid_lidar_bytes_per_launch_%dcorr" % n_corr),
                        "algorithmic_bytes_per_launch": bytes_alg}}
    try:
        # PCIe-inclusive: the handle API a Ceres EvaluationCallback uses (adapters/slam_residuals_hip.h) -- poses in,
        # residuals and both Jacobians out to host memory, every evaluation.  Never the figure above.
        h_corr = corr.cpu().numpy()
        h_off = (np.arange(n_blocks + 1, dtype=np.int64) * n_per).astype(np.int32)
        h_bs, h_bt, h_poses = bs.cpu().numpy(), bt.cpu().numpy(), poses.cpu().numpy()
        hnd = C.c_void_p()
        _lib.check(lib.nhip_resid_batch_create(0, _lib.ptr(h_corr), _lib.ptr(h_off), _lib.ptr(h_bs), _lib.ptr(h_bt),
                                               n_blocks, 1000, C.byref(hnd)))
        h_r = np.empty(2 * n_corr)
        h_js, h_jt = np.empty(6 * n_corr), np.empty(6 * n_corr)
        _lib.check(lib.nhip_resid_batch_eval(hnd, _lib.ptr(h_poses), _lib.ptr(h_r), _lib.ptr(h_js), _lib.ptr(h_jt)))
        t0 = time.perf_counter()
        _lib.check(lib.nhip_resid_batch_eval(hnd, _lib.ptr(h_poses), _lib.ptr(h_r), _lib.ptr(h_js), _lib.ptr(h_jt)))
        dt_j = time.perf_counter() - t0
        t0 = time.perf_counter()
        _lib.check(lib.nhip_resid_batch_eval(hnd, _lib.ptr(h_poses), _lib.ptr(h_r), None, None))
        dt_r = time.perf_counter() - t0
        lib.nhip_resid_batch_free(hnd)
        out["host_buffer_api"] = {
            "seconds_per_eval_with_jacobians": dt_j, "seconds_per_eval_residuals_only": dt_r,
            "correspondences_per_s_with_jacobians": n_corr / dt_j,
            "bytes_to_host_with_jacobians": 112.0 * n_corr,
            "same_result_as_device_api": bool(np.array_equal(h_js, js.cpu().numpy())),
            "note": "nhip_resid_batch_eval: 24 KB of poses up, 112 B per correspondence down over PCIe into pageable "
                    "host memory; the per-block normal equations (icp_front_half) move 224 B per BLOCK instead"}
        del h_r, h_js, h_jt
    except Exception as e:
        out["host_buffer_api_error"] = repr(e)
    if with_cpu:
        # CPU side of the same blocks: the oracle's Jet<6> autodiff restatement (what
        # ceres::AutoDiffCostFunction does per block), blocks across OpenMP threads like Ceres' num_threads.
        from oracle import oracle as O
        k = min(n_blocks, 8 * O.num_threads())
        h_corr = corr[:k * n_per].cpu().numpy()
        h_off = np.arange(k + 1, dtype=np.int32) * n_per
        h_bs, h_bt, h_poses = bs[:k].cpu().numpy(), bt[:k].cpu().numpy(), poses.cpu().numpy()
        t0 = time.perf_counter()
        wr, w0, w1 = O.lidar_batch(0, h_corr, h_off, h_bs, h_bt, h_poses, True, O.num_threads())
        dt = time.perf_counter() - t0
        ok = bool(np.allclose(res[:2 * k * n_per].cpu().numpy(), wr, rtol=1e-9, atol=1e-9) and
                  np.allclose(js[:6 * k * n_per].cpu().numpy().reshape(-1, 3), w0, rtol=1e-9, atol=1e-8))
        out["cpu_baseline"] = {"value": k * n_per / dt, "unit": "correspondences/s", "cores": O.num_threads(), "kind": "port",
                               "sample": "%d blocks x %d, Jet<6> autodiff restatement, OpenMP over blocks" % (k, n_per),
                               "gpu_matches_oracle_on_sample": ok}
    return out


def bench_icp(bag, xy, off, with_cpu, window=10, iters=5):
    """SURVEY 8f rows 1-2 at BASELINE configs[2] scale: 1000 poses, window 10 -> 9,945 (i, j) blocks of
    1081-point scans: correspondence search (K5) + per-block normal equations, all resident in HBM."""
    import torch
    from nautilus_amd import _lib
    from nautilus_amd.correspondence import IcpBatch, window_pairs
    lib = _lib.load()
    nrm = np.concatenate(bag.normals).astype(np.float32)
    bs, bt = window_pairs(bag.n_scans, window)
    batch = IcpBatch(xy, nrm, off, bs, bt)
    batch.set_poses(bag.odom)
    n_corr = batch.search()
    torch.cuda.synchronize()
    lib.nhip_timing_reset()
    lib.nhip_timing_enable(1)
    for _ in range(iters):
        batch.search(sync=False)
        batch.normal_equations(_lib.NHIP_LIDAR_POINT)
    torch.cuda.synchronize()
    lib.nhip_timing_enable(0)
    ms, n = C.c_double(0), C.c_int32(0)
    _lib.check(lib.nhip_timing_get(_lib.NHIP_TIMER_CORR, C.byref(ms), C.byref(n)))
    t_search = ms.value / max(n.value, 1)
    _lib.check(lib.nhip_timing_get(_lib.NHIP_TIMER_NORMEQ, C.byref(ms), C.byref(n)))
    t_neq = ms.value / max(n.value, 1)
    cand = float(np.sum((off[bs + 1] - off[bs]).astype(np.float64) * (off[bt + 1] - off[bt])))
    pts_in = float(np.sum(off[bs + 1] - off[bs]) + np.sum(off[bt + 1] - off[bt]))
    out = {"workload": "configs[2] shape: %d blocks (window %d) of 1081-point scans" % (len(bs), window),
           "correspondences": n_corr, "corr_search_ms": t_search,
           "corr_search_blocks_per_s": len(bs) / (t_search * 1e-3),
           "corr_search_exhaustive_equivalent_candidates_per_s": cand / (t_search * 1e-3),
           "corr_search_roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "traffic": None,
                                    "achieved": (8.0 * pts_in + 32.0 * n_corr) / (t_search * 1e-3) / 1e9,
                                    "frac": (8.0 * pts_in + 32.0 * n_corr) / (t_search * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "note": "8 B per source and target point read + 32 B per kept row written; the "
                                            "kernel is bound by its LDS sort and divergent bucket walks, not by HBM"},
           "normal_eq_ms": t_neq, "normal_eq_correspondences_per_s": n_corr / (t_neq * 1e-3),
           "normal_eq_roofline": {"bound": "hbm", "achieved": 32.0 * n_corr / (t_neq * 1e-3) / 1e9,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": 32.0 * n_corr / (t_neq * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                  "note": "32 B read per correspondence, 224 B written per block"}}
    if with_cpu:
        from oracle import oracle as O
        k = min(len(bs), 4 * O.num_threads())
        t0 = time.perf_counter()
        O.corr_search_batch(xy, nrm, off, bs[:k], bt[:k], O.pose_affines(bag.odom), 0.25, O.num_threads())
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": k / dt, "unit": "blocks/s", "cores": O.num_threads(), "kind": "port",
                               "sample": "%d blocks, oracle linear-scan restatement, OpenMP" % k,
                               "gpu_blocks_per_s": len(bs) / (t_search * 1e-3)}
    return out


if __name__ == "__main__":
    main()
