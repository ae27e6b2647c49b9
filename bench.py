#!/usr/bin/env python
"""bench.py -- PatchMatch MVS throughput on MI355X (BASELINE.json metric:
"PatchMatch Mpix/s @2560x1920", config[1]: 100-image ring, S=20, photometric only).

One process per GPU (torch.distributed / RCCL only for the barrier and the
max-over-ranks reduction: the path shards by reference image, no data-path
collective -- SURVEY.md section 8e). A "step" is one batch of `--batch` reference images
solved concurrently on this rank's GPU (each: upload from HBM-resident inputs,
reference filter, 2x2 footprint packing, initial cost, 5 x 4 sweeps, extraction).

Prints ONE JSON line (rank 0). See DESIGN.md "Measurement" for the derivation of
roofline.achieved (algorithmic bytes per sweep launch) and the CPU baseline sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=2560)
    ap.add_argument("--height", type=int, default=1920)
    ap.add_argument("--num-src", type=int, default=20)
    ap.add_argument("--ring", type=int, default=100, help="cameras on the full ring (config[1]: 100)")
    ap.add_argument("--batch", type=int, default=16, help="reference images solved concurrently per step")
    ap.add_argument("--cols", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--groups", type=int, default=1,
                    help="batches per step, each on its own stream (their sweep launches overlap)")
    ap.add_argument("--no-image-cache", action="store_true",
                    help="pack every problem's source images privately instead of sharing them")
    ap.add_argument("--no-ba", action="store_true", help="skip the secondary bundle-adjustment measurement")
    ap.add_argument("--ba-frames", type=int, default=1000)
    ap.add_argument("--ba-points", type=int, default=200000)
    ap.add_argument("--ba-track", type=int, default=10)
    ap.add_argument("--ba-iters", type=int, default=10)
    ap.add_argument("--ba2-frames", type=int, default=5000, help="BASELINE config[4] on one GPU (secondary.ba2)")
    ap.add_argument("--ba2-points", type=int, default=2000000)
    ap.add_argument("--no-ba2", action="store_true")
    ap.add_argument("--cpu-crop", type=str, default="512x384")
    ap.add_argument("--reference-build", action="store_true",
                    help="also time the reference's own kernel (oracle/_ref/libref_pm.so, its .cu files compiled for gfx950 "
                         "with a software texture fetch) on the CPU baseline's crop, one iteration = 4 sweeps; minutes")
    ap.add_argument("--no-fusion", action="store_true", help="skip the stereo-fusion leg (needs the geometric leg)")
    ap.add_argument("--no-dropin", action="store_true",
                    help="skip the seam leg (one problem at a time / one host thread per problem, as the reference's controller calls it)")
    ap.add_argument("--no-geom", action="store_true",
                    help="skip the geometric-consistency leg (BASELINE config[2]'s two-pass flow at 2560x1920)")
    return ap.parse_args()


def cpu_baseline(views, ref, src, dmin, dmax, crop_wh):
    """Oracle (CPU restatement, all host cores) on a bounded sample of the same workload:
    a centre crop of one full-resolution reference image against its full-resolution
    sources, all 20 sweeps. Per-pixel work is that of the full image."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pm_oracle
    cw, ch = crop_wh
    v = views[ref]
    H, W = v.gray.shape
    x0, y0 = (W - cw) // 2, (H - ch) // 2
    K = v.K.copy()
    K[0, 2] -= x0
    K[1, 2] -= y0
    imgs = []
    for i, vv in enumerate(views):
        if i == ref:
            imgs.append(dict(K=K, R=vv.R, T=vv.T, gray=np.ascontiguousarray(vv.gray[y0:y0 + ch, x0:x0 + cw])))
        else:
            imgs.append(dict(K=vv.K, R=vv.R, T=vv.T, gray=vv.gray))
    o = pm_oracle.default_options(depth_min=dmin, depth_max=dmax, geom_consistency=0, filter=1)
    o.num_threads = os.cpu_count() or 0   # every host core (importing torch caps OpenMP's default at the physical cores)
    t = time.time()
    pm_oracle.run(o, imgs, ref, src)
    dt = time.time() - t
    return dict(value=cw * ch / 1e6 / dt, unit="Mpix/s", cores=int(pm_oracle.lib().pmo_num_threads()),
                kind="port",
                sample=f"oracle/pm_oracle.c on a {cw}x{ch} centre crop of one {W}x{H} reference image, "
                       f"S={len(src)} full-resolution sources, 5x4 sweeps, {dt:.1f} s wall")


def ba_secondary(a, local_rank, with_cpu, rank=0, world=1, dev=None):
    """BASELINE.json config[3]: global BA, 1000 cameras x 200k points, SIMPLE_RADIAL, track length 10
    (benchmark/runtime/bundle_adjustment.cc noise model), Schur-PCG: LM iterations per second over
    the first `--ba-iters` iterations (linearise + Schur-Jacobi + PCG + step evaluation, inputs
    resident in HBM), with the fp64 CPU oracle timed on the same problem at N = 1.
    N > 1: the SAME problem solved by all ranks together (strong scaling), observations sharded
    over the GPUs, partial sums combined by RCCL all-reduces on the solver's stream inside
    ba_solve_sharded -- once sharded by image (what BASELINE.json names) and once by point (only
    camera-space vectors travel); the time of a solve is the max over ranks. Called by every rank."""
    import ctypes as C
    from colmap_amd import distributed as D, estimators as est, scene
    from colmap_amd._lib import lib
    d = scene.synthesize_flat(a.ba_frames, a.ba_points, a.ba_track, seed=42,
                              noise=scene.SyntheticNoiseOptions(0.01, 1.0, 0.05, 1.0))
    fp = est.FlatProblem.from_arrays(d)
    est.fix_gauge_two_cams(fp)
    so = est.SolverOptions(max_num_iterations=a.ba_iters)
    n_obs = len(fp.obs_pose)
    sharded = {}
    if world == 1:
        est.solve_flat(fp.copy(), est.SolverOptions(max_num_iterations=2), gpu_index=local_rank)  # warm-up
        s = est.solve_flat(fp.copy(), so, gpu_index=local_rank)
        seconds = s.lm_seconds
        parallelism = "one GPU"
        n_obs_rank = n_obs
    else:
        comm = est.Communicator("rccl", gpu_index=local_rank)
        best = None
        for name, mode in (("image", est.SHARD_BY_IMAGE), ("point", est.SHARD_BY_POINT)):
            comm.sharding = mode
            est.solve_flat(fp.copy(), est.SolverOptions(max_num_iterations=2), gpu_index=local_rank, comm=comm)  # warm-up
            si = est.solve_flat(fp.copy(), so, gpu_index=local_rank, comm=comm)
            ti = D.max_over_ranks(si.lm_seconds, dev)
            sharded[name] = {"LM_iterations_per_s": si.num_iterations / ti, "lm_iterations": si.num_iterations,
                             "pcg_iterations": int(si.total_linear_iterations), "final_cost": si.final_cost,
                             "observations_on_rank0": est.shard_num_observations(fp, 0, world, mode)}
            if best is None or ti < best[1]:
                best = (si, ti, name, mode)
        s, seconds, best_name, best_mode = best
        # leave the timing of the best mode in the library's per-thread counters
        comm.sharding = best_mode
        s = est.solve_flat(fp.copy(), so, gpu_index=local_rank, comm=comm)
        seconds = D.max_over_ranks(s.lm_seconds, dev)
        comm.close()
        parallelism = (f"observations sharded by {best_name} over {world} GPUs, RCCL all-reduce on the solver's "
                       f"stream (both shardings measured, see `sharded`)")
        n_obs_rank = est.shard_num_observations(fp, rank, world, best_mode)
    ms, n, _ = C.c_double(), C.c_int64(), C.c_int64()
    lib().ba_last_spmv_timing(C.byref(ms), C.byref(n), C.byref(_))
    mfma_ms, mfma_n = C.c_double(), C.c_int64()
    lib().ba_last_mfma_timing(C.byref(mfma_ms), C.byref(mfma_n))
    alg = 352 * n_obs_rank  # 2 passes over the fp64 Jacobian rows: 2 x 2 x (6 + 2 + 3) x 8 B per observation
    avg_ms = ms.value / max(n.value, 1)
    out = {
        "metric": "BA LM-iters/s @1000 imgs",
        "value": s.num_iterations / seconds,
        "unit": "LM-iterations/s",
        "n_gpus": world,
        "scaling": "strong",
        "dtype": "f64",
        "config": {"workload": f"global BA, {a.ba_frames} cameras x {a.ba_points} points, SIMPLE_RADIAL, "
                               f"track length {a.ba_track} ({n_obs} observations), gauge TWO_CAMS_FROM_WORLD, "
                               f"implicit Schur PCG + Schur-Jacobi, first {a.ba_iters} LM iterations",
                   "lm_iterations": s.num_iterations, "pcg_iterations": int(s.total_linear_iterations),
                   "cost": [s.initial_cost, s.final_cost], "parallelism": parallelism},
        "roofline": {"bound": "hbm", "kernel": "implicit Schur product (ba_obs_jx + ba_point_pass + ba_block_jtv)",
                     "achieved": alg / (avg_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg / (avg_ms * 1e-3) / 1e9 / 8000.0, "traffic": ba_pmc_traffic(n_obs_rank),
                     # dense camera-block contraction (Schur-Jacobi blocks M_b = sum J^T (I - G) J on
                     # v_mfma_f64_16x16x4_f64): share of the LM time, and its flop rate
                     "mfma_time_frac": (mfma_ms.value * 1e-3) / max(seconds, 1e-12),
                     "mfma_avg_launch_ms": mfma_ms.value / max(mfma_n.value, 1),
                     "algorithmic_bytes_per_launch": alg, "avg_launch_ms": avg_ms, "launches_timed": int(n.value)},
    }
    # whole LM iteration against SURVEY.md section 8(d)'s byte model N_o x (256 + 176 k), k = PCG iterations
    # per LM iteration (the model assumes fp32 Jacobian storage; `fp64_storage` re-costs it for the fp64
    # arrays this solver keeps): linearise + Schur preparation + k implicit products + step evaluation
    k_pcg = s.total_linear_iterations / max(s.num_iterations, 1)
    t_lm = seconds / max(s.num_iterations, 1)
    model = n_obs_rank * (256.0 + 176.0 * k_pcg)
    out["roofline"]["lm_iteration"] = {
        "pcg_iterations_per_lm_iteration": k_pcg, "ms_per_lm_iteration": t_lm * 1e3,
        "model_bytes": model, "achieved": model / t_lm / 1e9, "frac": model / t_lm / 1e9 / 8000.0,
        "fp64_storage": {"model_bytes": 2 * model, "frac": 2 * model / t_lm / 1e9 / 8000.0}}
    if world == 1:
        # The reference's own default tier at this size (SPARSE_SCHUR for 51..1000 images,
        # bundle_adjustment_ceres.cc:203-213): exact Newton steps from the explicitly formed reduced camera system,
        # blocked Cholesky on the f64 matrix cores (colmap_amd/csrc/ba_schur_explicit.hip). Reported beside the
        # benchmarked Schur-PCG tier: which one is faster per LM iteration, and how far each has come.
        n_c = int(est.num_camera_parameters(fp))
        est.solve_flat(fp.copy(), est.SolverOptions(max_num_iterations=1, linear_solver_type=est.SOLVER_SPARSE_SCHUR),
                       gpu_index=local_rank)  # warm-up (first launches of the tier's kernels), like the iterative leg's
        se = est.solve_flat(fp.copy(), est.SolverOptions(max_num_iterations=min(a.ba_iters, 6),
                                                         linear_solver_type=est.SOLVER_SPARSE_SCHUR), gpu_index=local_rank)
        out["exact_tier"] = {
            "linear_solver": "SPARSE_SCHUR (explicit reduced camera system, dense in HBM, pair-major formation, blocked f64-MFMA Cholesky)",
            "LM_iterations_per_s": se.num_iterations / max(se.lm_seconds, 1e-12), "lm_iterations": se.num_iterations,
            "cost": [se.initial_cost, se.final_cost], "tier_used": se.linear_solver_used,
            "mfma_time_frac": se.factor_seconds / max(se.lm_seconds, 1e-12),
            # n_c^3 / 3 flop per factorisation of the reduced camera system (one per LM iteration) over the time
            # inside the blocked Cholesky; n_c = variable camera parameters (6 per pose + intrinsics - gauge)
            "cholesky_tflops": n_c ** 3 / 3.0 * se.num_iterations / max(se.factor_seconds, 1e-12) / 1e12,
            "reduced_system_size": n_c,
            "faster_tier_per_lm_iteration": "ITERATIVE_SCHUR (PCG)" if out["value"] > se.num_iterations / max(se.lm_seconds, 1e-12)
                                            else "SPARSE_SCHUR"}
        # the same solve with the point-major formation (one atomic per term, the round-3 kernel): the difference in
        # seconds per LM iteration is the formation's (the factorisation is the same code; `factor_ms_per_lm_iteration`)
        lib().colmap_amd_set_switch(b"COLMAP_AMD_BA_FORM_PAIRS", b"0")   # development switch (csrc/switches.h)
        try:
            sp = est.solve_flat(fp.copy(), est.SolverOptions(max_num_iterations=min(a.ba_iters, 6),
                                                             linear_solver_type=est.SOLVER_SPARSE_SCHUR), gpu_index=local_rank)
        finally:
            lib().colmap_amd_set_switch(b"COLMAP_AMD_BA_FORM_PAIRS", None)
        out["exact_tier"]["formation"] = {
            "pair_major_ms_per_lm_iteration": 1e3 * se.lm_seconds / max(se.num_iterations, 1),
            "point_major_ms_per_lm_iteration": 1e3 * sp.lm_seconds / max(sp.num_iterations, 1),
            "factor_ms_per_lm_iteration": 1e3 * se.factor_seconds / max(se.num_iterations, 1),
            "same_cost_log": bool(len(sp.log_cost) == len(se.log_cost) and
                                  np.allclose(sp.log_cost, se.log_cost, rtol=1e-9, atol=0.0))}
    if world == 1:
        # what the drop-in adapter does with DEFAULT options at this size: AUTO resolved on the image count with the
        # backend's measured GPU thresholds (estimators.resolve_linear_solver)
        tier = est.resolve_linear_solver(a.ba_frames)
        names = {est.SOLVER_ITERATIVE_SCHUR: "ITERATIVE_SCHUR", est.SOLVER_DENSE_SCHUR: "DENSE_SCHUR", est.SOLVER_SPARSE_SCHUR: "SPARSE_SCHUR"}
        out["default_route"] = {
            "what": "Mi355xBundleAdjuster / estimators.BundleAdjuster with default options (linear_solver_type = AUTO)",
            "tier": names[tier],
            "LM_iterations_per_s": out["value"] if tier == est.SOLVER_ITERATIVE_SCHUR else out["exact_tier"]["LM_iterations_per_s"]}
        # SURVEY.md 8(d)'s second BA-1 variant: ONE camera shared by all images (num_rigs = 1) -- the case the
        # reference's FAQ warns about (doc/faq.rst:626-632: shared intrinsics densify the Schur complement): the single
        # intrinsics block is reached from every observation, the Schur-Jacobi cross terms carry real load
        d1 = dict(d)
        d1["obs_cam"] = np.zeros_like(d["obs_cam"])
        d1["cams"] = d["cams"][:1].copy()
        d1["cam_model"] = d["cam_model"][:1].copy()
        fp1 = est.FlatProblem.from_arrays(d1)
        est.fix_gauge_two_cams(fp1)
        est.solve_flat(fp1.copy(), est.SolverOptions(max_num_iterations=2), gpu_index=local_rank)  # warm-up
        s1 = est.solve_flat(fp1.copy(), so, gpu_index=local_rank)
        e1 = est.solve_flat(fp1.copy(), est.SolverOptions(max_num_iterations=min(a.ba_iters, 6),
                                                          linear_solver_type=est.SOLVER_SPARSE_SCHUR), gpu_index=local_rank)
        out["shared_intrinsics"] = {
            "workload": f"the same {a.ba_frames} images x {a.ba_points} points with ONE SIMPLE_RADIAL camera shared by all images",
            "iterative": {"LM_iterations_per_s": s1.num_iterations / max(s1.lm_seconds, 1e-12), "lm_iterations": s1.num_iterations,
                          "pcg_iterations": int(s1.total_linear_iterations), "cost": [s1.initial_cost, s1.final_cost]},
            "exact": {"LM_iterations_per_s": e1.num_iterations / max(e1.lm_seconds, 1e-12), "lm_iterations": e1.num_iterations,
                      "cost": [e1.initial_cost, e1.final_cost], "reduced_system_size": int(est.num_camera_parameters(fp1))}}
        if with_cpu:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import ba_oracle
            c1 = fp1.copy()
            o1 = est.solve_flat(c1, est.SolverOptions(max_num_iterations=9), solve_fn=ba_oracle.solve_fn, num_threads=os.cpu_count() or 0)
            h1 = est.solve_flat(fp1.copy(), est.SolverOptions(max_num_iterations=9), gpu_index=local_rank)
            m1 = min(len(o1.log_cost), len(h1.log_cost))
            out["shared_intrinsics"]["hip_vs_oracle_max_rel_cost_diff"] = \
                float(np.max(np.abs(h1.log_cost[:m1] - o1.log_cost[:m1]) / o1.log_cost[:m1])) if m1 else None
            out["shared_intrinsics"]["hip_vs_oracle_same_pcg_iterations"] = bool(
                np.array_equal(h1.log_linear_iters[:m1], o1.log_linear_iters[:m1]))
            out["shared_intrinsics"]["oracle_LM_iterations_per_s"] = o1.num_iterations / max(o1.lm_seconds, 1e-12)
    if world == 1:
        # what a caller pays besides the LM loop (the mapper calls BA hundreds of times: set-up is product time), and the
        # same problem solved to the reference's own stopping rule instead of the first `--ba-iters` iterations
        import time as _time
        out["setup_seconds"] = s.setup_seconds
        out["whole_solve_LM_iterations_per_s"] = s.num_iterations / max(s.lm_seconds + s.setup_seconds, 1e-12)
        t0 = _time.time()
        sc_ = est.solve_flat(fp.copy(), est.SolverOptions(), gpu_index=local_rank)   # COLMAP's defaults: <= 100 iterations, gradient 1e-4
        out["to_convergence"] = {
            "options": "defaults (max_num_iterations 100, gradient_tolerance 1e-4, function / parameter tolerance 0)",
            "termination": sc_.termination_type.name, "lm_iterations": sc_.num_iterations,
            "successful_steps": sc_.num_successful_steps, "pcg_iterations": int(sc_.total_linear_iterations),
            "lm_seconds": sc_.lm_seconds, "setup_seconds": sc_.setup_seconds, "wall_seconds": _time.time() - t0,
            "LM_iterations_per_s": sc_.num_iterations / max(sc_.lm_seconds, 1e-12),
            "cost": [sc_.initial_cost, sc_.final_cost]}
        if not a.no_ba2:
            out["ba2"] = ba2_leg(a, local_rank, with_cpu)
    if sharded:
        out["sharded"] = sharded
    if with_cpu and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import ba_oracle
        c = fp.copy()
        sc = est.solve_flat(c, est.SolverOptions(max_num_iterations=9), solve_fn=ba_oracle.solve_fn,
                            num_threads=os.cpu_count() or 0)  # ~10 s on the box's host cores (all of them)
        # parity on the benchmark problem itself: the HIP cost log of the same 9 iterations against the oracle's
        sh = est.solve_flat(fp.copy(), est.SolverOptions(max_num_iterations=9), gpu_index=local_rank)
        m = min(len(sc.log_cost), len(sh.log_cost))
        rel = float(np.max(np.abs(sh.log_cost[:m] - sc.log_cost[:m]) / sc.log_cost[:m])) if m else None
        out["cpu_baseline"] = dict(value=sc.num_iterations / sc.lm_seconds, unit="LM-iterations/s",
                                   cores=int(ba_oracle.lib().bao_num_threads()), kind="port",
                                   sample=f"oracle/ba_oracle.c (fp64, OpenMP), first {sc.num_iterations} LM iterations "
                                          f"of the same problem, {sc.lm_seconds:.1f} s",
                                   hip_vs_oracle_max_rel_cost_diff=rel, hip_vs_oracle_iterations_compared=m,
                                   hip_vs_oracle_same_pcg_iterations=bool(
                                       np.array_equal(sh.log_linear_iters[:m], sc.log_linear_iters[:m])))
    return out


def ba2_leg(a, local_rank, with_cpu):
    """BASELINE.json config[4] on ONE GPU (it fits: 20 M observations x ~0.7 KB): 5000 cameras x 2 M points, track
    length 10, half PINHOLE / half SIMPLE_RADIAL (benchmark/runtime/bundle_adjustment.cc:40-81's generator), Schur-PCG,
    first `--ba-iters` LM iterations; bytes resident on the device during the solve; parity of the first cost values
    against the fp64 oracle at a tenth of the size through the same code path (the oracle needs minutes at full size)."""
    import ctypes as C
    import threading
    import time as _time
    from colmap_amd import estimators as est, scene
    frames, points, track = a.ba2_frames, a.ba2_points, a.ba_track
    noise = scene.SyntheticNoiseOptions(0.01, 1.0, 0.05, 1.0)
    t0 = _time.time()
    d = scene.synthesize_flat(frames, points, track, seed=44, mixed_models=True, noise=noise)
    fp = est.FlatProblem.from_arrays(d)
    est.fix_gauge_two_cams(fp)
    t_gen = _time.time() - t0
    models = {int(m): int((fp.cam_model == m).sum()) for m in np.unique(fp.cam_model)}
    free0 = torch.cuda.mem_get_info(local_rank)[0]
    low = [free0]
    stop = threading.Event()

    def watch():   # bytes resident during the solve = the lowest free-memory reading while it runs
        while not stop.is_set():
            low[0] = min(low[0], torch.cuda.mem_get_info(local_rank)[0])
            _time.sleep(0.02)
    th = threading.Thread(target=watch, daemon=True)
    th.start()
    try:
        s = est.solve_flat(fp.copy(), est.SolverOptions(max_num_iterations=a.ba_iters), gpu_index=local_rank)
    finally:
        stop.set()
        th.join()
    n_obs = len(fp.obs_pose)
    out = {"workload": f"global BA, {frames} cameras x {points} points, track length {track} ({n_obs} observations), "
                       f"camera models {models} (CameraModelId: count), gauge TWO_CAMS_FROM_WORLD, implicit Schur PCG, "
                       f"first {a.ba_iters} LM iterations, one GPU",
           "LM_iterations_per_s": s.num_iterations / max(s.lm_seconds, 1e-12), "lm_iterations": s.num_iterations,
           "pcg_iterations": int(s.total_linear_iterations), "ms_per_lm_iteration": 1e3 * s.lm_seconds / max(s.num_iterations, 1),
           "setup_seconds": s.setup_seconds, "cost": [s.initial_cost, s.final_cost],
           "device_bytes_resident": int(free0 - low[0]), "bytes_per_observation": (free0 - low[0]) / max(n_obs, 1),
           "host_generation_seconds": t_gen,
           "roofline_lm_iteration_fp64_storage_frac":
               2 * n_obs * (256.0 + 176.0 * s.total_linear_iterations / max(s.num_iterations, 1)) /
               (s.lm_seconds / max(s.num_iterations, 1)) / 1e9 / 8000.0}
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import ba_oracle
        d10 = scene.synthesize_flat(frames // 10, points // 10, track, seed=44, mixed_models=True, noise=noise)
        f10 = est.FlatProblem.from_arrays(d10)
        est.fix_gauge_two_cams(f10)
        o10 = est.solve_flat(f10.copy(), est.SolverOptions(max_num_iterations=3), solve_fn=ba_oracle.solve_fn,
                             num_threads=os.cpu_count() or 0)
        h10 = est.solve_flat(f10.copy(), est.SolverOptions(max_num_iterations=3), gpu_index=local_rank)
        m = min(len(o10.log_cost), len(h10.log_cost))
        out["parity_at_one_tenth"] = {
            "workload": f"{frames // 10} cameras x {points // 10} points, same generator and code path",
            "hip_vs_oracle_max_rel_cost_diff": float(np.max(np.abs(h10.log_cost[:m] - o10.log_cost[:m]) / o10.log_cost[:m])) if m else None,
            "iterations_compared": m,
            "same_pcg_iterations": bool(np.array_equal(h10.log_linear_iters[:m], o10.log_linear_iters[:m]))}
    return out


def ba_pmc_traffic(n_obs):
    """FETCH_SIZE + WRITE_SIZE of the three kernels of one implicit Schur product (scripts/profile_ba.sh,
    separate --pmc passes; profiles/ba_schur_traffic.json), or None when the committed passes were taken
    on another problem size."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "ba_schur_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if int(t["observations"]) != int(n_obs):
            return None
        return float(t["fetch_bytes_per_product"]) + float(t["write_bytes_per_product"])
    except (OSError, KeyError, ValueError):
        return None


def pmc_traffic(images_per_launch, kernel):
    """HBM-side bytes per sweep launch from the committed rocprofv3 PMC passes (FETCH_SIZE +
    WRITE_SIZE of the sweep kernel, scripts/profile_pm.sh; counters cannot be read from inside
    this process). None unless the passes were taken on THIS kernel, image layout and launch shape."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pm_sweep_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if int(t["images_per_launch"]) != int(images_per_launch) or t.get("kernel") != kernel \
                or t.get("layout") != PM_IMAGE_LAYOUT:
            return None
        return float(t["fetch_bytes_per_launch"]) + float(t["write_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


# packed source-image layout of the library this script measures (pm_internal.h: kFpStrip); a traffic file taken on
# another layout does not describe this build
PM_IMAGE_LAYOUT = "strips16x2-dword-footprints"


def reference_build_leg(views, ref, src, dmin, dmax, crop_wh):
    """The reference's OWN PatchMatchCuda on this GPU: oracle/_ref/libref_pm.so = patch_match_cuda.cu / gpu_mat_prng.cu /
    gpu_mat_ref_image.cu compiled for gfx950 where they lie (make -C oracle ref; only the texture fetch is a software
    stand-in, gfx950 has no image instructions). Baseline leg only (checker code, like cpu_baseline): the same crop of one
    reference image against its full-resolution sources, num_iterations = 1 (four sweeps, one per direction; the
    reference has no smaller unit), scaled to the five iterations of a solve."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pm_oracle
    import ref_pm
    if not ref_pm.available():
        return {"error": "oracle/_ref/libref_pm.so not built (make -C oracle ref needs /root/reference)"}
    cw, ch = crop_wh
    v = views[ref]
    H, W = v.gray.shape
    x0, y0 = (W - cw) // 2, (H - ch) // 2
    K = v.K.copy()
    K[0, 2] -= x0
    K[1, 2] -= y0
    imgs = [dict(K=K, R=vv.R, T=vv.T, gray=np.ascontiguousarray(vv.gray[y0:y0 + ch, x0:x0 + cw])) if i == ref
            else dict(K=vv.K, R=vv.R, T=vv.T, gray=vv.gray) for i, vv in enumerate(views)]
    o = pm_oracle.default_options(depth_min=dmin, depth_max=dmax, geom_consistency=0, filter=1, num_iterations=1)
    r = ref_pm.RefPatchMatch(o, imgs, ref, src)
    t = time.time()
    out = r.run()
    dt = time.time() - t
    r.close()
    return {"Mpix_per_s": cw * ch / 1e6 / (5.0 * dt), "seconds_for_4_sweeps": dt, "kept_by_filter": float((out["depth"] > 0).mean()),
            "sample": f"{cw}x{ch} centre crop of one {W}x{H} reference image, S={len(src)} full-resolution sources, "
                      f"num_iterations=1 timed, x5 for the 5x4 sweeps of a solve",
            "note": "the reference's kernel as written (one thread per image column, 32-thread blocks, per-thread state in "
                    "scratch) with a software texture stand-in: a same-node figure for the reference's SOURCE, not for "
                    "COLMAP on an NVIDIA GPU"}


def dropin_leg(a, problem, batched_value):
    """The same problems through the seam exactly as the reference drives it: its
    controller keeps ONE problem in flight per entry of --PatchMatchStereo.gpu_index (mvs/patch_match.cc:177,
    190-204); listing a GPU several times gives that many worker threads on it (:375-383). Measured:
      one_at_a_time    one PatchMatchCuda-shaped handle after the other (pm_create / pm_run / pm_get_* / pm_destroy): a
                       single 2560 x 1920 problem has 1920 .. 2560 columns for 5120 resident waves -- the library then
                       runs one column per wave group and a helper wave per column (pm_sweep_pair_kernel);
      threads          `--batch` host threads, each with its own handle and stream on this GPU (the repeated-index
                       route), all running at once;
      batched          pm_run_batch of `--batch` problems (the primary number above: what the controller-side
                       batching patch of INTEGRATION.md gives)."""
    import threading
    from colmap_amd import mvs
    n1 = 2
    t_one, one_kernel = 0.0, ""
    for j in range(n1):   # one handle alive at a time, as in the reference's worker thread (mvs/patch_match.cc:394-440)
        pm = problem(j)[0]
        torch.cuda.synchronize()
        t0 = time.time()
        pm.Run()
        pm.GetDepthMap()
        torch.cuda.synchronize()
        t_one += (time.time() - t0) / n1
        one_kernel = pm.GetSweepKernelName()
        pm.close()
    nt = a.batch
    pms = [problem(j)[0] for j in range(nt)]
    errs = []

    def work(pm):
        try:
            pm.Run()
            pm.GetDepthMap()
        except Exception as e:  # noqa: BLE001 (reported in the line)
            errs.append(repr(e))
    torch.cuda.synchronize()
    t0 = time.time()
    th = [threading.Thread(target=work, args=(pm,)) for pm in pms]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    t_thr = time.time() - t0
    for pm in pms:
        pm.close()
    mpix = a.width * a.height / 1e6
    out = {"one_at_a_time_Mpix_per_s": mpix / t_one, "one_at_a_time_kernel": one_kernel, "threads": nt, "threads_Mpix_per_s": nt * mpix / t_thr,
           "batched_Mpix_per_s": batched_value,
           "note": "create + run + depth-map read-back per problem, image cache shared; the controller of the "
                   "reference calls it this way (one problem per worker thread)"}
    if errs:
        out["errors"] = errs[:3]
    return out


def geometric_leg(a, views, images, cache, local_rank):
    """BASELINE.json config[2]'s pass at config[1]'s size: PatchMatch with the geometric consistency
    term and both filters at 2560x1920, S = 20, through the reference's two-pass flow
    (mvs/patch_match.cc:183-204): a photometric pass (no filter) for every image a geometric problem
    uses as a source, its depth / normal maps kept in HBM (device-to-device, what
    PatchMatchController.Run does between the passes), then ONE batch of `--batch` reference images with
    geom_consistency = filter = true. Timed: the geometric batch (create + 5 x 4 sweeps + extraction),
    and the whole two-pass job. Sources of a view near the end of this rank's window are its 20 nearest
    views inside the window."""
    from colmap_amd import mvs
    S, nb = a.num_src, a.batch
    half = S // 2
    n_need = min(len(views), nb + 2 * half)   # references half .. half+nb-1 and their sources

    def nearest(i, n):
        lo = max(0, min(i - half, n - 1 - S))
        return [j for j in range(lo, lo + S + 1) if j != i][:S]

    def options(i, geom):
        dmin, dmax = views[i][4] * 0.9, views[i][5] * 1.1
        return mvs.PatchMatchOptions(gpu_index=str(local_rank), depth_min=dmin, depth_max=dmax, sigma_spatial=5.0,
                                     geom_consistency=geom, filter=geom, columns_per_group=a.cols,
                                     threads_per_group=a.threads)
    torch.cuda.synchronize()
    t0 = time.time()
    maps = [None] * len(views)
    for b0 in range(0, n_need, nb):
        idx = list(range(b0, min(b0 + nb, n_need)))
        pms = [mvs.PatchMatch(options(i, False), mvs.PatchMatch.Problem(i, nearest(i, len(views)), images), cache)
               for i in idx]
        mvs.run_batch(pms, wait=True)
        for i, pm in zip(idx, pms):
            maps[i] = pm.GetDeviceMaps()
            pm.close()
    torch.cuda.synchronize()
    t_photo = time.time() - t0
    depth_maps = [None if m is None else m[0] for m in maps]
    normal_maps = [None if m is None else m[1:] for m in maps]
    refs = list(range(half, half + nb))
    t1 = time.time()
    pms = [mvs.PatchMatch(options(i, True),
                          mvs.PatchMatch.Problem(i, [i + o for o in range(-half, half + 1) if o != 0][:S], images,
                                                 depth_maps, normal_maps), cache) for i in refs]
    mvs.run_batch(pms, wait=True)
    torch.cuda.synchronize()
    t_geom = time.time() - t1
    ms, n = pms[0].GetSweepTiming()
    ev = [pm.GetEvaluationCount() for pm in pms]
    nf = 0 if a.no_fusion else min(8, nb)   # filtered maps handed to the fusion leg (host arrays: its ABI takes host data)
    fmaps = [(refs[k], pms[k].GetDepthMap(), pms[k].GetNormalMap()) for k in range(nf)]
    kept = float(np.mean([(pm.GetDepthMap() > 0).mean() for pm in pms[:2]]))
    for pm in pms:
        pm.close()
    pix = a.width * a.height
    return fmaps, {
        "metric": "PatchMatch Mpix/s @2560x1920, geometric consistency pass",
        "value": nb * pix / 1e6 / t_geom, "unit": "Mpix/s",
        "config": {"workload": f"geom_consistency=true, filter=true (photometric + geometric filters), {a.width}x{a.height}, "
                               f"S={S}, {nb} reference images in one batch; source depth / normal maps = this run's "
                               f"photometric pass, resident in HBM"},
        "seconds": t_geom, "avg_sweep_launch_ms": ms / max(n, 1), "launches_timed": n,
        "ncc_evaluations_per_pixel_per_sweep": sum(e[0] for e in ev) / (nb * pix * 20.0),
        "pixels_kept_by_filter": kept,
        "two_pass": {"photometric_images": n_need, "photometric_seconds": t_photo,
                     "Mpix_per_s_both_passes": nb * pix / 1e6 / (t_photo * nb / n_need + t_geom),
                     "note": "photometric seconds charged to the geometric batch in proportion nb / photometric_images "
                             "(in a whole workspace every image is solved once per pass)"},
    }


def fusion_leg(a, views, fmaps, with_cpu=True):
    """Stereo fusion (reference mvs/fusion.cc) of the geometric leg's filtered depth /
    normal maps: 8 neighbouring 2560x1920 images, default StereoFusionOptions, every image overlapping
    every other. `value` = pixels / the time on the device (passes, medians, compaction, read-back of the points:
    fusion_last_timing); the host-to-HBM upload of the maps the ABI takes as host arrays is reported beside it.
    Turn order = the reference's pool schedule, one thread per ten-row stripe.
    cpu_baseline: the oracle in the reference's sequential order (mode 0: one thread, as mvs/fusion.cc runs with
    num_threads = 1) on the first four of the same images."""
    import ctypes as C
    from colmap_amd import fusion
    from colmap_amd._lib import lib
    imgs = []
    for (i, depth, normal) in fmaps:
        K, R, T, g = views[i][0], views[i][1], views[i][2], views[i][3]
        gray = g.cpu().numpy()
        imgs.append(fusion.FusionImage(a.width, a.height, K, R, T, np.stack([gray, gray, gray], -1), depth, normal))
    n = len(imgs)
    overlap = [[j for j in range(n) if j != i] for i in range(n)]
    opt = fusion.StereoFusionOptions()
    # warm-up like every other leg's: the first fusion_run of a process loads the kernels' code object and sizes hipCUB's
    # scratch (230 ms on the box, scripts/fusion_setup_timing.py) -- two small images, then the timed call
    small = [fusion.FusionImage(im.width, im.height, im.K, im.R, im.T, None, np.ascontiguousarray(im.depth_map[:96, :128]),
                                np.ascontiguousarray(im.normal_map[:, :96, :128])) for im in imgs[:2]]
    for im in small:
        im.width, im.height = 128, 96
    fusion.fuse(opt, small, [[1], [0]])
    t = time.time()
    pts = fusion.fuse(opt, imgs, overlap)
    dt = time.time() - t
    st = [C.c_int64() for _ in range(4)]
    lib().fusion_last_stats(*[C.byref(x) for x in st])
    images_, seeds, rounds, walks = [x.value for x in st]
    up, dev = C.c_double(), C.c_double()
    lib().fusion_last_timing(C.byref(up), C.byref(dev))
    mpix = n * a.width * a.height / 1e6
    # bytes the algorithm has to move once: depth 4 + normal 12 + colour 3 per pixel in, mask stamp 4 + claim word 8
    # read-modify-write, ~40 bytes per fused point out
    alg = n * a.width * a.height * (4 + 12 + 3 + 2 * 4 + 2 * 8) + 40 * len(pts.xyz)
    # value = END TO END through the C ABI (host depth / normal / colour maps in, host points out): fusion_run takes host
    # arrays, so a caller always pays the upload; the device-only rate is reported beside it
    out = {"metric": "stereo fusion Mpix/s @2560x1920", "value": mpix / dt, "unit": "Mpix/s",
           "config": {"workload": f"StereoFusion defaults, {n} images {a.width}x{a.height} (geometric leg's filtered maps), "
                                  f"all-to-all overlap; end to end through fusion_run (host maps in, host points out)"},
           "seconds": {"device": dev.value, "upload_and_setup": up.value, "end_to_end": dt},
           "device_only_Mpix_per_s": mpix / max(dev.value, 1e-9),
           "fused_points": int(len(pts.xyz)), "seed_pixels": int(seeds),
           "passes_per_image": rounds / max(images_, 1), "walks_per_seed_pixel": walks / max(seeds, 1),
           "roofline": {"bound": "latency: one sequential wave per pool thread (ten-row stripe) walks its turns in the "
                                 "reference's order; a wave's absorbed pixels x the round trip of a node's dependent "
                                 "loads (mask word, depth, normal), not bytes, set the time",
                        "achieved": alg / max(dev.value, 1e-9) / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": alg / max(dev.value, 1e-9) / 1e9 / 8000.0, "algorithmic_bytes": alg,
                        "ms_per_pass": dev.value * 1e3 / max(rounds, 1),
                        "pool_threads": (a.height + 9) // 10, "traffic": None}}
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import fusion_oracle
        ov = [[j for j in range(n) if j != i] for i in range(n)]
        # the reference's own multi-threaded path (mvs/fusion.cc:247-269: ThreadPool(num_threads), default all cores;
        # oracle mode 3 = real threads racing on the masks as the reference's do) on the SAME images, end to end
        cores = os.cpu_count() or 1
        t = time.time()
        ref_mt = fusion_oracle.fuse(opt, imgs, ov, mode=3)
        dmt = time.time() - t
        m = min(4, n)
        sub = imgs[:m]
        t = time.time()
        ref = fusion_oracle.fuse(opt, sub, [[j for j in range(m) if j != i] for i in range(m)], mode=0)
        dc = time.time() - t
        out["cpu_baseline"] = {"value": mpix / dmt, "unit": "Mpix/s", "cores": cores, "kind": "port",
                               "sample": f"oracle/fusion_oracle.cpp mode 3 (the reference's thread pool, num_threads = all "
                                         f"{cores} cores, stripes of ten rows pulled from a shared counter) on the same {n} "
                                         f"images, {len(ref_mt.xyz)} points, {dmt:.2f} s wall",
                               "one_core": {"value": m * a.width * a.height / 1e6 / dc, "unit": "Mpix/s", "cores": 1,
                                            "sample": f"mode 0 (row-major, one thread) on the first {m} images, "
                                                      f"{len(ref.xyz)} points"}}
    return out


def spawn_command(gpus, argv, port, script=None):
    """Command line and environment of the self-spawned multi-GPU run (spawn_ranks): one rank per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve), dmabuf IPC for RCCL."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return cmd, env


def check_world(gpus, world):
    """The launcher must have started exactly --gpus ranks (the driver's contract: `--gpus N` under
    torch.distributed.run with --nproc-per-node N)."""
    if world != gpus:
        raise SystemExit(f"bench.py: --gpus {gpus} but the launcher started WORLD_SIZE={world} ranks")


def spawn_ranks(a):
    """`python bench.py --gpus N` (N > 1) without a launcher around it: become the launcher. Re-executes
    this script under torch.distributed.run with one rank per GPU (RCCL over xGMI); rank 0 of that job
    prints the single JSON line, `n_gpus` in it is the world size RCCL saw."""
    import socket
    n_dev = torch.cuda.device_count()
    if n_dev < a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} requested but only {n_dev} GPU(s) are visible")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd, env = spawn_command(a.gpus, sys.argv[1:], port)
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    check_world(a.gpus, world)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from colmap_amd import distributed as D, mvs, synthetic as syn

    S = a.num_src
    nsteps = a.steps + a.warmup
    nref = nsteps * a.batch * a.groups
    # this rank's window of the ring: `nref` consecutive reference cameras + S/2 neighbours either side
    half = S // 2
    step_deg = 360.0 / a.ring
    # disjoint reference images per rank (weak scaling)
    idx0, nviews = D.rank_window(rank, nref, half)
    cams = syn.ring_cameras(nviews, a.width, a.height, 2400.0 * a.width / 2560.0, arc_deg=step_deg * (nviews - 1),
                            start_deg=idx0 * step_deg)
    views = []
    for (K, R, T) in cams:
        g, d, n = syn.render_view(K, R, T, a.width, a.height, seed=0, device=str(dev))
        views.append((K, R, T, g, float(d.min()), float(d.max())))
    images = [mvs.Image(K, R, T, g) for (K, R, T, g, _, _) in views]

    # packed source images are shared between the problems that use them (as the reference's
    # CachedWorkspace shares bitmaps on the host); --no-image-cache packs them per problem
    cache = None if a.no_image_cache else mvs.ImageCache(local_rank)

    def problem(j):  # j-th reference image of this rank
        ref = half + j
        src = [ref + o for o in range(-half, half + 1) if o != 0][:S]
        dmin, dmax = views[ref][4] * 0.9, views[ref][5] * 1.1
        opt = mvs.PatchMatchOptions(gpu_index=str(local_rank), depth_min=dmin, depth_max=dmax,
                                    sigma_spatial=5.0, geom_consistency=False, filter=True,
                                    columns_per_group=a.cols, threads_per_group=a.threads)
        return mvs.PatchMatch(opt, mvs.PatchMatch.Problem(ref, src, images), cache), (ref, src, dmin, dmax)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    sweep_ms, sweep_n = 0.0, 0
    evals_sweep, evals_init = 0, 0   # NCC evaluations executed in the timed steps (this rank)
    pms_keepalive = []
    kernel_names = set()   # sweep kernels the timed steps launched (the library reports what it ran)
    launch_shape = [a.batch, 1]   # reference images per sweep launch, launches in flight together (pm_get_launch_shape)

    def run_step(step, record):
        nonlocal sweep_ms, sweep_n, evals_sweep, evals_init
        grps = [[problem((step * a.groups + g) * a.batch + b)[0] for b in range(a.batch)]
                for g in range(a.groups)]
        for pms in grps:
            for pm in pms:
                pm.Create()
        for pms in grps:
            mvs.run_batch(pms, wait=False)  # one launch per sweep covers the whole batch
        for pms in grps:
            for pm in pms:
                pm.Synchronize()
        if record:
            kernel_names.add(grps[0][0].GetSweepKernelName())
            ms, n = grps[0][0].GetSweepTiming()
            sweep_ms += ms
            sweep_n += n
            launch_shape[:] = grps[0][0].GetLaunchShape()
            for pms in grps:
                for pm in pms:
                    a_, b_ = pm.GetEvaluationCount()
                    evals_sweep += a_
                    evals_init += b_
        for pms in grps:
            for pm in pms:
                pm.close()

    for w in range(a.warmup):
        run_step(w, False)
    barrier()
    t0 = time.time()
    for k in range(a.steps):
        run_step(a.warmup + k, True)
    barrier()
    dt = time.time() - t0
    dt = D.max_over_ranks(dt, dev)
    images_done = sum(D.gather_counts(a.steps * a.batch * a.groups, dev))

    if rank == 0:
        pix_per_image = a.width * a.height
        total_images = images_done
        value = total_images * pix_per_image / 1e6 / dt
        # (40 + 24*S) algorithmic HBM bytes per pixel per sweep launch
        # A launch sweeps one sub-batch (pm_run_batch runs 16+ problems as two sub-batches on two streams: the drain of
        # one sweep launch is filled by the other's); `avg_ms` = HIP events around the launches of the first sub-batch,
        # during which `conc` launches share the GPU. achieved = the bytes the GPU sweeps in that time.
        ipl, conc = launch_shape[0], launch_shape[1] * a.groups
        alg_bytes = (40 + 24 * S) * pix_per_image * ipl
        avg_ms = sweep_ms / max(sweep_n, 1)
        achieved = conc * alg_bytes / (avg_ms * 1e-3) / 1e9
        kernel_name = " + ".join(sorted(kernel_names))
        out = {
            "metric": "PatchMatch Mpix/s @2560×1920",
            "value": value,
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"PatchMatch MVS, photometric only (geom_consistency=false, filter=true), "
                            f"{a.ring}-camera ring stand-in for Gerrard-Hall, {a.width}x{a.height}, "
                            f"S={S} sources, window 11x11, 15 samples, 5x4 sweeps; "
                            f"{a.batch} reference images per step per GPU",
                "images_per_step_per_gpu": a.batch,
                "shared_source_images": not a.no_image_cache,
                "parallelism": f"reference images sharded over {world} GPU(s), no data-path collective",
            },
            # The kernel has no dense contraction. Counters (profiles/r04_pm_pmc_diagnosis.json, r04_pm_pmc_summary.json):
            # the VALU is active 86-96 % of the launch AND the texture-address path 78 % -- both pipes are close to
            # full at once. `bound` says so, `achieved / peak / frac` keep the HBM figures BASELINE.json asks for
            # (algorithmic bytes / launch time against 8 TB/s), the VALU-side figures follow.
            "roofline": {
                "bound": "valu-fp32 issue and texture-address (gather) rate, both > 78 % busy (HBM fraction reported as BASELINE.json asks)",
                "kernel": kernel_name,
                "achieved": achieved,
                "peak": 8000.0,
                "unit": "GB/s",
                "frac": achieved / 8000.0,
                "traffic": pmc_traffic(ipl, kernel_name),
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": avg_ms,
                "launches_timed": sweep_n,
                "images_per_launch": ipl,
                "concurrent_launches": conc,
                "achieved_one_launch_alone": alg_bytes / (avg_ms * 1e-3) / 1e9,
                # what the kernel is actually bound by: NCC evaluations counted
                # in-kernel (identical hypothesis / view pairs of a pixel are evaluated once) x 121
                # taps x ~30 flop per tap against the 157.3 TFLOP/s fp32 vector peak; rank 0's counts
                "taps_per_s": (evals_sweep + evals_init) * 121 / dt,
                "valu_frac": (evals_sweep + evals_init) * 121 * 30.0 / dt / 157.3e12,
                "ncc_evaluations_per_pixel_per_sweep": evals_sweep / max(a.steps * a.batch * a.groups * pix_per_image * 20, 1),
                "note": "kernel without a dense contraction: by PMC the VALU is active 86 % of the launch and the "
                        "texture-address unit 78 % (gathers of 64 scattered dwords, ~40 L1 accesses each); "
                        "taps_per_s / valu_frac carry the useful arithmetic; the HBM fraction is reported because "
                        "BASELINE.json asks for it. traffic = FETCH_SIZE + WRITE_SIZE per launch from "
                        "profiles/pm_sweep_traffic.json, null unless that file was measured on this kernel and "
                        "image layout (DESIGN.md 1.5)",
            },
        }
        # the CPU baseline and the secondary (single-GPU) BA measurement belong to the N = 1 run only
        if not a.no_cpu_baseline and world == 1:
            ref, src, dmin, dmax = problem(0)[1]
            cw, ch = [int(x) for x in a.cpu_crop.split("x")]
            host_views = [syn.View(K, R, T, g.cpu().numpy(), None, None) for (K, R, T, g, _, _) in views]
            out["cpu_baseline"] = cpu_baseline(host_views, ref, src, dmin, dmax, (cw, ch))
            if a.reference_build:
                out["reference_hip_build"] = reference_build_leg(host_views, ref, src, dmin, dmax, (cw, ch))
        if not a.no_dropin and world == 1:
            out["dropin"] = dropin_leg(a, problem, value)
        if not a.no_geom and world == 1:
            fmaps, out["geometric"] = geometric_leg(a, views, images, cache, local_rank)
            if fmaps:
                mvs.release_cached_memory()
                out["fusion"] = fusion_leg(a, views, fmaps, not a.no_cpu_baseline)
        if not a.no_ba and world == 1:
            del pms_keepalive[:]
            torch.cuda.empty_cache()
            mvs.release_cached_memory()
            out["secondary"] = ba_secondary(a, local_rank, not a.no_cpu_baseline)
    # N > 1: the secondary (bundle adjustment) leg is ONE solve sharded over all ranks
    if not a.no_ba and world > 1:
        torch.cuda.empty_cache()
        mvs.release_cached_memory()
        # The sharded BA leg is the only part of this script with data-path collectives. The primary
        # measurement above must not be lost to it: if the leg fails on any rank or does not finish in time,
        # rank 0 prints the line without it and every rank leaves (each rank runs the same watchdog).
        import threading

        def give_up(reason):
            if rank == 0:
                out["secondary"] = {"error": reason}
                print(json.dumps(out), flush=True)
            os._exit(0)
        watchdog = threading.Timer(float(os.environ.get("COLMAP_AMD_BENCH_BA_TIMEOUT", "240")),
                                   give_up, args=("sharded bundle-adjustment leg timed out",))
        watchdog.daemon = True
        watchdog.start()
        try:
            sec = ba_secondary(a, local_rank, False, rank, world, dev)
        except Exception as e:  # the other ranks are stuck in a collective now: their watchdogs release them
            give_up(f"sharded bundle-adjustment leg failed on rank {rank}: {e!r}")
        watchdog.cancel()
        if rank == 0:
            out["secondary"] = sec
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
