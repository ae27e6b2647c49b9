"""Time-to-cost of the three linear-solver tiers on this GPU at several problem sizes (torch-free; ~1 min on the box).

The reference keeps separate solver-tier thresholds for its CPU and GPU (Ceres-CUDA) solvers
(estimators/bundle_adjustment_ceres.h:68-71: 50 / 1000 images on the CPU, 200 / 4000 on the GPU). The MI355X backend's
AUTO rule (colmap_amd/estimators.py: resolve_linear_solver, include/colmap_amd/bundle_adjustment.hpp) is set from THIS
table: for every size, the seconds each tier needs to bring the cost to within 1e-4 (relative) of the best final cost any
tier reaches in 30 LM iterations (None: not reached), with the LM time spread evenly over a solve's iterations.

    gpurun -- 'python scripts/ba_tier_crossover.py > gpurun_out/ba_tier_crossover.json'
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from colmap_amd import estimators as est, scene

SIZES = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "50,100,200,500,1000,2000".split(","))]
TIERS = [("DENSE_SCHUR", est.SOLVER_DENSE_SCHUR), ("SPARSE_SCHUR", est.SOLVER_SPARSE_SCHUR), ("ITERATIVE_SCHUR", est.SOLVER_ITERATIVE_SCHUR)]
rows = []
for frames in SIZES:
    d = scene.synthesize_flat(frames, 200 * frames, 10, seed=42, noise=scene.SyntheticNoiseOptions(0.01, 1.0, 0.05, 1.0))
    fp = est.FlatProblem.from_arrays(d)
    est.fix_gauge_two_cams(fp)
    runs = {}
    for name, tier in TIERS:
        if tier == est.SOLVER_DENSE_SCHUR and frames > 1000:
            continue
        so = est.SolverOptions(max_num_iterations=30, linear_solver_type=tier)
        est.solve_flat(fp.copy(), est.SolverOptions(max_num_iterations=1, linear_solver_type=tier), gpu_index=0)   # warm-up
        s = est.solve_flat(fp.copy(), so, gpu_index=0)
        runs[name] = s
    target = min(s.final_cost for s in runs.values()) * (1 + 1e-4)
    row = {"images": frames, "points": 200 * frames, "observations": int(len(fp.obs_pose)),
           "n_c": int(est.num_camera_parameters(fp)), "target_cost": target, "tiers": {}}
    for name, s in runs.items():
        log = np.asarray(s.log_cost)
        reached = bool((log <= target).any())
        hit = int(np.argmax(log <= target)) if reached else None   # log[k] = cost after k iterations
        per_it = s.lm_seconds / max(s.num_iterations, 1)
        row["tiers"][name] = {"iterations_to_target": hit, "lm_iterations": int(s.num_iterations),
                              "ms_per_lm_iteration": 1e3 * per_it,
                              "ms_to_target": 1e3 * per_it * max(hit, 1) if reached else None,
                              "final_cost": float(s.final_cost), "tier_used": int(s.linear_solver_used)}
    row["fastest"] = min(row["tiers"], key=lambda k: row["tiers"][k]["ms_to_target"] if row["tiers"][k]["ms_to_target"] is not None else 1e30)
    rows.append(row)
    print(json.dumps(row), file=sys.stderr, flush=True)
print(json.dumps({"rows": rows}, indent=1))
