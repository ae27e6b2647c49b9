"""Which linear-solver tier brings a bundle adjustment down fastest on this GPU, by problem size (torch-free).

The reference keeps separate solver-tier thresholds for its CPU and GPU (Ceres-CUDA) solvers
(estimators/bundle_adjustment_ceres.h:68-71: 50 / 1000 images on the CPU, 200 / 4000 on the GPU). The MI355X backend's
AUTO rule (colmap_amd/estimators.py: resolve_linear_solver, include/colmap_amd/bundle_adjustment.hpp) is set from THIS
table. Criterion (round 6; the round-5 one -- "within 1e-4 of the best final cost of any tier" -- flipped with the seed):

    target = the cost the EXACT tier has reached after 3 LM steps (three exact Newton-type steps from the benchmark's
             noise level are where its curve flattens);
    exact  = 3 x its seconds per LM iteration;
    PCG    = (LM iterations the Schur-PCG tier needs to get to the target or below, of at most 30) x its seconds per
             LM iteration, None when it does not get there.

Every size is solved for several seeds of the benchmark's generator (200 points per image, track length 10, its noise
model); the table reports the per-seed times and their medians, and the rule that follows: the largest size up to
which the exact tier's median wins, with every larger size going to PCG (monotone by construction).

    gpurun -- 'python scripts/ba_tier_crossover.py [sizes] [seeds] > gpurun_out/ba_tier_crossover.json'
MEASUREMENT INFRASTRUCTURE."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from colmap_amd import estimators as est, scene

SIZES = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "50,100,200,350,500,700,1000,1500,2000,3000,4000").split(",")]
SEEDS = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "42,43,44").split(",")]
rows = []
for frames in SIZES:
    per_seed = []
    for seed in SEEDS:
        d = scene.synthesize_flat(frames, 200 * frames, 10, seed=seed, noise=scene.SyntheticNoiseOptions(0.01, 1.0, 0.05, 1.0))
        fp = est.FlatProblem.from_arrays(d)
        est.fix_gauge_two_cams(fp)
        n_c = int(est.num_camera_parameters(fp))
        res = {"seed": seed}
        ex = None
        if n_c <= 32768:
            tier = est.SOLVER_SPARSE_SCHUR
            est.solve_flat(fp.copy(), est.SolverOptions(max_num_iterations=1, linear_solver_type=tier), gpu_index=0)   # warm-up
            ex = est.solve_flat(fp.copy(), est.SolverOptions(max_num_iterations=3, linear_solver_type=tier), gpu_index=0)
            res["exact"] = {"ms_per_lm_iteration": 1e3 * ex.lm_seconds / max(ex.num_iterations, 1),
                            "ms_to_target": 1e3 * ex.lm_seconds, "cost_log": [float(c) for c in ex.log_cost],
                            "setup_ms": 1e3 * ex.setup_seconds, "tier_used": int(ex.linear_solver_used)}
        tier = est.SOLVER_ITERATIVE_SCHUR
        est.solve_flat(fp.copy(), est.SolverOptions(max_num_iterations=1, linear_solver_type=tier), gpu_index=0)       # warm-up
        it = est.solve_flat(fp.copy(), est.SolverOptions(max_num_iterations=30, linear_solver_type=tier), gpu_index=0)
        log = np.asarray(it.log_cost)
        per_it = it.lm_seconds / max(it.num_iterations, 1)
        res["pcg"] = {"ms_per_lm_iteration": 1e3 * per_it, "lm_iterations": int(it.num_iterations),
                      "pcg_iterations": int(it.total_linear_iterations), "setup_ms": 1e3 * it.setup_seconds,
                      "final_cost": float(it.final_cost)}
        if ex is not None and len(ex.log_cost):
            target = float(ex.log_cost[-1]) * (1.0 + 1e-9)
            reached = bool((log <= target).any())
            hit = int(np.argmax(log <= target)) + 1 if reached else None   # log[k] = cost after k + 1 iterations
            res["target_cost"] = target
            res["pcg"]["iterations_to_target"] = hit
            res["pcg"]["ms_to_target"] = 1e3 * per_it * hit if reached else None
        per_seed.append(res)
    row = {"images": frames, "points": 200 * frames, "n_c": n_c, "seeds": per_seed}
    ex_t = [r["exact"]["ms_to_target"] for r in per_seed if "exact" in r]
    pc_t = [r["pcg"].get("ms_to_target") for r in per_seed]
    row["exact_ms_to_target_median"] = float(np.median(ex_t)) if ex_t else None
    row["pcg_ms_to_target_median"] = float(np.median([t if t is not None else 1e30 for t in pc_t])) if ex_t else None
    if row["pcg_ms_to_target_median"] is not None and row["pcg_ms_to_target_median"] >= 1e29:
        row["pcg_ms_to_target_median"] = None
    row["pcg_reached_target_in_seeds"] = sum(t is not None for t in pc_t)
    e, p = row["exact_ms_to_target_median"], row["pcg_ms_to_target_median"]
    row["winner"] = "exact" if e is not None and (p is None or e <= p) else "pcg"
    rows.append(row)
    print(json.dumps({k: v for k, v in row.items() if k != "seeds"}), file=sys.stderr, flush=True)
# the monotone rule: exact up to the largest size below which exact wins everywhere
thr = 0
for r in rows:
    if r["winner"] == "exact":
        thr = r["images"]
    else:
        break
print(json.dumps({"criterion": "time to the exact tier's cost after 3 LM steps; medians over seeds " + str(SEEDS),
                  "rows": rows, "largest_size_with_exact_winning_from_the_bottom": thr}, indent=1))
