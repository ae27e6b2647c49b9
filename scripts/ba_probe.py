import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from colmap_amd import estimators as est, scene
import argparse
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=6); ap.add_argument("--points", type=int, default=40); ap.add_argument("--track", type=int, default=4)
ap.add_argument("--iters", type=int, default=100); ap.add_argument("--op32", type=int, default=0); ap.add_argument("--lst", type=int, default=0); ap.add_argument("--oracle", type=int, default=0); ap.add_argument("--gtol", type=float, default=1e-4)
ap.add_argument("--switch", action="append", default=[], help="development switch NAME=VALUE (csrc/switches.h), repeatable")
ap.add_argument("--shared", type=int, default=0, help="share this many cameras between all images (0: one camera per image)")
a = ap.parse_args()
for kv in a.switch:
    k, v = kv.split("=", 1); est.lib().colmap_amd_set_switch(k.encode(), v.encode())
t = time.time(); d = scene.synthesize_flat(a.frames, a.points, a.track, seed=42, noise=scene.SyntheticNoiseOptions(0.01, 1.0, 0.05, 1.0)); print("gen %.2fs" % (time.time() - t))
if a.shared > 0:
    d["obs_cam"] = (d["obs_cam"] % a.shared).astype(np.int32); d["cams"] = d["cams"][:a.shared].copy(); d["cam_model"] = d["cam_model"][:a.shared].copy()
fp = est.FlatProblem.from_arrays(d); est.fix_gauge_two_cams(fp)
so = est.SolverOptions(gradient_tolerance=a.gtol, max_num_iterations=a.iters, operator_precision=a.op32, linear_solver_type=a.lst)
for rep in range(2):
    b = fp.copy(); t = time.time(); s = est.solve_flat(b, so, gpu_index=0); dt = time.time() - t
    print(f"HIP rep{rep}: tier {s.linear_solver_used} factor {s.factor_seconds:.4f}s wall {dt:.3f}s setup {s.setup_seconds:.3f}s lm {s.lm_seconds:.3f}s iters {s.num_iterations} succ {s.num_successful_steps} pcg {s.total_linear_iterations} cost {s.initial_cost:.6e}->{s.final_cost:.6e} {s.termination_type.name} -> {s.num_iterations/max(s.lm_seconds,1e-9):.2f} LM-it/s")
if a.oracle:
    import ba_oracle
    b = fp.copy(); t = time.time(); s = est.solve_flat(b, so, solve_fn=ba_oracle.solve_fn); dt = time.time() - t
    print(f"oracle: wall {dt:.3f}s lm {s.lm_seconds:.3f}s iters {s.num_iterations} pcg {s.total_linear_iterations} cost ->{s.final_cost:.6e} -> {s.num_iterations/max(s.lm_seconds,1e-9):.2f} LM-it/s")
